"""CPU baseline worker (test / bench infrastructure, like the rest of oracle/): runs the
oracle's reference-literal float mode of config C3 on a slice of samples in its own
process with one BLAS thread, so bench.py can use several host cores the way an
embarrassingly parallel eval would (one sample stream per core).

  python -m oracle.cpu_baseline <payload.npz> <first> <last>   -> one JSON line

Only bench.py's cpu_baseline leg starts it; nothing in the product imports it."""
import json
import sys
import time


def main():
  path, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
  from threadpoolctl import threadpool_limits
  import numpy as np
  from oracle import snn_oracle as o
  with np.load(path) as z:
    bits = int(z["bits"])
    lb = [int(b) for b in z["layer_bits"]] if "layer_bits" in z.files else [bits] * 4
    which = {"conv0": 0, "conv1": 1, "conv2": 2, "dense": 3}
    def qw(prefix):
      quant = None
      if float(z[prefix + "_a"]) != -1.0:
        quant = {"kind": "duq", "bits": lb[which[prefix]], "a": float(z[prefix + "_a"]), "c": float(z[prefix + "_c"])}
      mask = z[prefix + "_mask"] if (prefix + "_mask") in z.files else None
      return o.QWeight(z[prefix + "_kernel"], quant, mask)
    cq = [qw("conv%d" % i) for i in range(3)]
    dq = qw("dense")
    bns = [dict(mean=z["bn%d_mean" % i], var=z["bn%d_var" % i], scale=z["bn%d_scale" % i],
                bias=z["bn%d_bias" % i]) for i in range(3)]
    x = z["x"][first:last].astype(np.float32)
  with threadpool_limits(limits=1):
    o.conv3_dense_forward(x[:1, :2], cq, bns, dq, mode="float")     # warm-up
    t0 = time.perf_counter()
    o.conv3_dense_forward(x, cq, bns, dq, mode="float")
    dt = time.perf_counter() - t0
  print(json.dumps({"samples": int(last - first), "seconds": dt}))


if __name__ == "__main__":
  main()
