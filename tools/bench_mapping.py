"""f1 probe: get_mapping_ind at the two scales the pipeline uses (8, 16) for an 8-frame 512^2 batch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, fresco_amd
g = synth.gen(0)
flows, occs = synth.make_flows(8, 512, g)
imgs = torch.rand(8, 3, 512, 512, generator=g)
fd, od, im = flows[1].cuda(), occs[1].cuda(), imgs.cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in (8.0, 16.0):
        fresco_amd.get_mapping_ind(fd, od, im, scale=s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("get_mapping_ind, N=8, 512^2, scales 8+16: %.3f ms on the GPU" % (dt * 1e3))
from oracle import fresco_oracle as O
t0 = time.perf_counter()
for s in (8.0, 16.0):
    O.mapping_ind(flows[1], occs[1], imgs, scale=s)
print("oracle (vectorised CPU restatement): %.1f ms; reference Python loop: ~900 ms on 8 CPU cores (SURVEY.md section 6)" % ((time.perf_counter() - t0) * 1e3))
