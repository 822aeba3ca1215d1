// Micro-benchmark: which lanes of a wave64 ds_read_b128 are served together?
// Each pattern maps lane -> 16-byte slot (mod 16 slots = the 64 banks); a pattern
// that is conflict-free under the true grouping runs at full LDS rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k(int *out, const int *slot, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384 / 4; i += 256) ((int *)lds)[i] = i;
  __syncthreads();
  int off = slot[lane] * 16 + lane * 256;   // one 256-byte bank row per lane: equal slots conflict
  v4i acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      v4i v = *(v4i *)(lds + ((off & ~255) | ((off + j * 16) & 255)));
      acc += v;
    }
    asm volatile("" : "+v"(off));   // opaque: the reads stay in the loop
  }
  out[blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  int *out, *dslot;
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&dslot, 64 * 4);
  struct P { const char *name; int (*f)(int); };
  P ps[] = {
      {"linear lane%16 (H1: 16 consecutive)", [](int l) { return l % 16; }},
      {"all same slot (broadcast)", [](int l) { return 0; }},
      {"lane%8 (pairs 2-way under any 16-group)", [](int l) { return l % 8; }},
      {"H2 {0-7,32-39}: l%8 + 8*(l>>5)", [](int l) { return l % 8 + 8 * (l >> 5); }},
      {"H3 {0-7,16-23}: l%8 + 8*((l>>4)&1)", [](int l) { return l % 8 + 8 * ((l >> 4) & 1); }},
      {"H4 {0-7,8-15}: l%16", [](int l) { return l % 16; }},
      {"H5 quads {0-3,16-19,32-35,48-51}: l%4 + 4*(l>>4)", [](int l) { return l % 4 + 4 * (l >> 4); }},
      {"H6 {0-3,8-11,...}: l%4 + 4*((l>>3)&3)", [](int l) { return l % 4 + 4 * ((l >> 3) & 3); }},
      {"8 lanes distinct, rest same: min(l,8)%16", [](int l) { return l < 8 ? l : 8; }},
      {"lanes 0-31 distinct halves: l%32/2", [](int l) { return (l % 32) / 2; }},
      {"stride 2 slots: (2l)%16", [](int l) { return (2 * l) % 16; }},
  };
  for (auto &p : ps) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = p.f(l);
    (void)hipMemcpy(dslot, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, dslot, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    // per CU: 4 waves x iters x 16 reads x 1024 B
    const double bytes = 4.0 * iters * 16 * 1024;
    printf("%-52s %.3f ms  %.1f B/clk/CU @2.4GHz\n", p.name, ms, bytes / (ms * 1e-3 * 2.4e9));
  }
  return 0;
}
