import sys, torch, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from dualhyp_amd import ops
from dualhyp_amd.synth import uniform, stream_id
D="cuda:0"
def U(shape, b, name): return uniform(shape, b, stream_id(5, name)).to(D)
d, I = 256, 384
x = U((4, d), 1.0, "x"); W = U((512, d), 0.2, "w"); A = U((48, d), 0.1, "a"); B = U((512, 16), 0.2, "b")
w1, w2 = U((I, d), 0.2, "w1"), U((I, d), 0.2, "w2"); wp = U((d, I), 0.2, "wp"); res = U((4, d), 1.0, "r")
for M in (1, 2, 4):
    xs = x[:M].contiguous()
    xa = ops.linear(xs, A)
    y = ops.linear(xs, W, epilogue=ops.EPI_LORA, xa=xa, lora_b=B, lora_scale=2.0, splits=(256, 384))
    act = ops.linear(xs, w1, epilogue=ops.EPI_SWIGLU, w2=w2)
    o = ops.linear(act, wp, resid=res[:M].contiguous())
    if M == 1: ref = (xa.clone(), y.clone(), act.clone(), o.clone())
    else:
        print(M, [ (a[0] != b[0]).sum().item() for a, b in zip(ref, (xa, y, act, o))])
# attention decode n_seq 1 vs 2
H, G, hs, S = 4, 2, 64, 128
kc = U((2, G, S, hs), 1.0, "kc"); vt = U((2, G, hs, S), 1.0, "vt"); q = U((2, H, hs), 1.0, "q")
i32 = torch.int32
y1 = ops.attn_decode(q[:1].contiguous(), kc, vt, torch.tensor([0], dtype=i32, device=D), torch.tensor([30], dtype=i32, device=D))
y2 = ops.attn_decode(q, kc, vt, torch.tensor([0, 1], dtype=i32, device=D), torch.tensor([30, 23], dtype=i32, device=D))
print("attn decode row0 differs:", (y1[0] != y2[0]).sum().item())
# rmsnorm
w = U((d,), 1.0, "nw")
n1 = ops.rmsnorm(x[:1].contiguous(), w, 1e-5, row_tail=torch.ones(1, dtype=torch.uint8, device=D))
n2 = ops.rmsnorm(x[:2].contiguous(), w, 1e-5, row_tail=torch.ones(2, dtype=torch.uint8, device=D))
print("rmsnorm row0 differs:", (n1[0] != n2[0]).sum().item())
