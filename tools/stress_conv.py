"""Randomised shapes through the fused conv block (tests/stress.py): the MFMA kernels against
the direct-form kernel on the same inputs, bit for bit.  usage: stress_conv.py [N] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.stress import conv_block_random, dense_block_random

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = conv_block_random(torch.device("cuda:0"), n, seed, verbose=True)
bad += dense_block_random(torch.device("cuda:0"), n, seed + 1, verbose=True)
print("failures:", len(bad), bad)
sys.exit(1 if bad else 0)
