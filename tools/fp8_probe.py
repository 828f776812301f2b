"""Probe (GPU box): dh_quant_rows_fp8 / dh_linear_fp8 against the oracle on random data; prints where they differ."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops
from dualhyp_amd.synth import uniform, stream_id
from oracle import ger_oracle as O

for rows, K in ((5, 512), (33, 4096)):
    x = uniform((rows, K), 3.0, stream_id(77, f"q{K}"))
    q_ref, s_ref = O.quantize_rows_fp8(x)
    q, s = ops.quant_rows_fp8(x.cuda())
    q, s = q.cpu(), s.cpu()
    bad = (q != q_ref.view(torch.uint8)).nonzero()
    print(f"rows {rows} K {K}: scale equal {torch.equal(s, s_ref.view(-1))}; {bad.size(0)} of {q.numel()} bytes differ")
    for r, c in bad[:6].tolist():
        inv = 448.0 / x[r].float().abs().max()
        print(f"   [{r},{c}] x={float(x[r,c])!r} x*inv={float(x[r,c].float()*inv)!r} gpu 0x{int(q[r,c]):02x} torch 0x{int(q_ref.view(torch.uint8)[r,c]):02x}")
M, N, K = 7, 512, 512
x, w = uniform((M, K), 1.5, stream_id(77, "fx")), uniform((N, K), 0.05, stream_id(77, "fw"))
wq, ws = O.quantize_rows_fp8(w)
xq_ref, xs_ref = O.quantize_rows_fp8(x)
y_ref = O.linear_fp8(x, wq, ws.view(-1)).float()
y = ops.linear_fp8(xq_ref.view(torch.uint8).cuda(), xs_ref.view(-1).cuda(), wq.view(torch.uint8).cuda(), ws.view(-1).cuda()).float().cpu()
acc = xq_ref.float() @ wq.float().T
print("with the oracle's xq: max |y - y_ref| / rms", ((y - y_ref).abs().max() / y_ref.pow(2).mean().sqrt()).item())
ones_m, ones_n = torch.ones(M).cuda(), torch.ones(N).cuda()
y1 = ops.linear_fp8(xq_ref.view(torch.uint8).cuda(), ones_m, wq.view(torch.uint8).cuda(), ones_n).float().cpu()
print("unit scales: gpu acc vs cpu acc (bf16-rounded) max rel", ((y1 - acc.bfloat16().float()).abs().max() / acc.abs().max()).item())
print("acc sample", acc[0, :4].tolist(), y1[0, :4].tolist())
