import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests.stress import conv_block_random, dense_block_random
dev = torch.device("cuda:0")
f1 = conv_block_random(dev, 150, 777)
print("conv failures:", f1)
f2 = dense_block_random(dev, 150, 778)
print("dense failures:", f2)
