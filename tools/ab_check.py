"""Digest of dense_nn results on a bench-sized bfloat16 batch, to compare development builds of the library
(ISPA_HIP_LIBRARY=...) with the in-tree one:  python tools/ab_check.py [pairs grid channels k [f32]]
Prints SHA-256 of indices and distances, and the exact-path statistics."""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
import pose_estimation_amd as pea

b, g, c, k = (int(x) for x in (sys.argv[1:5] + ["512", "64", "128", "1"][len(sys.argv[1:5]):]))
tdt = torch.float32 if sys.argv[5:6] == ["f32"] else torch.bfloat16
gen = torch.Generator(device="cuda").manual_seed(99)
def feats():
    x = torch.randn(b, c, g, g, device="cuda", generator=gen)
    return (x / x.norm(dim=1, keepdim=True)).to(tdt).contiguous()
q, r = feats(), feats()
r[:, :, 0, :7] = r[:, :, 1, :7]          # some exact ties
dist, ind, st = pea.correspondence.dense_nn(q, r, k, return_stats=True)
torch.cuda.synchronize()
print("ind", hashlib.sha256(ind.cpu().numpy().tobytes()).hexdigest()[:16],
      "dist", hashlib.sha256(dist.cpu().numpy().tobytes()).hexdigest()[:16], st)
