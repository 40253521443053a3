import sys, json; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/open-diffusiongs_amd')
import torch, bench
print(json.dumps(bench.scene_512(torch.device('cuda:0'))))
