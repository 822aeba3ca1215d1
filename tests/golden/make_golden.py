"""Regenerates the committed fixtures under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference cannot run in this environment (jax / flax are not installed,
SURVEY.md section 8c), so these vectors are outputs of oracle/snn_oracle.py on
the seeded inputs of tests/cases.py -- they pin the oracle against drift and
travel to the GPU box; they are not outputs of the reference itself.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import snn_oracle as o  # noqa: E402
from tests import cases  # noqa: E402


def main():
  out_dir = os.path.dirname(os.path.abspath(__file__))
  total = 0
  for name, fn in cases.GOLDEN.items():
    exp = fn(o)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **exp)
    total += os.path.getsize(path)
    print("%-28s %8d B  %s" % (name, os.path.getsize(path),
                               ", ".join(sorted(exp))[:90]))
  print("total %d B" % total)


if __name__ == "__main__":
  main()
