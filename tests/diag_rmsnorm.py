import torch, sys
sys.path.insert(0, '.')
from dualhyp_amd import ops
from dualhyp_amd.synth import uniform, stream_id
from oracle import ger_oracle as O
def rbf(v): return v.bfloat16().float()
d=2048; rows=37
x = uniform((rows,d), 2.0, stream_id(11,f"x{d}"))
w = (1 + uniform((d,), 0.25, stream_id(11,f"w{d}")).float()).bfloat16()
got = ops.rmsnorm(x.cuda(), w.cuda(), 1e-5).float().cpu()
xf = x.float(); eps=1e-5
sq = rbf(xf*xf); ms = rbf(sq.sum(-1,keepdim=True)/d); t = rbf(ms+eps); r = rbf(1/torch.sqrt(t)); xn = rbf(xf*r); f = rbf(w.float()*xn)
o = O.rmsnorm(x, w, eps).float()
print("gpu vs formula rows differing:", (got!=f).any(-1).sum().item(), "elems", (got!=f).sum().item())
print("gpu vs torch-cpu rows differing:", (got!=o).any(-1).sum().item(), "elems", (got!=o).sum().item())
print("formula vs torch-cpu rows differing:", (f!=o).any(-1).sum().item())
# where does torch-cpu deviate from the formula?
ms_t = torch.mean(x*x, dim=-1, keepdim=True).float()
print("ms torch vs formula:", (ms_t!=ms).sum().item())
r_t = torch.rsqrt((torch.mean(x*x, dim=-1, keepdim=True)+eps)).float()
print("r torch vs formula:", (r_t!=r).sum().item())
print(torch.__config__.show().split('\n')[0:8])
import subprocess; print(subprocess.run("lscpu | grep -E 'Model name|Flags' | cut -c1-400", shell=True, capture_output=True, text=True).stdout)
