"""BASELINE configs[2] (training targets, batch 16, 100 GT boxes per image): the kernel times and the wall-clock
training halves that bench.py reports as `training_targets_config3`.   python tools/time_config3.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
print(json.dumps(bench.targets_microbench(bench.Path(dev), dev), indent=1))
