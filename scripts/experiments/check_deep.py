"""Deep-pipelined 256x256 NT kernel vs the 8-wave 256x256 kernel on the same inputs (expected: bit-identical, the
k-order of the fp32 accumulation is the same).  Usage: python scripts/check_deep.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops

dev = "cuda:0"
os.environ["DRN_NT_TILE"] = "256"


def run(dt, B, Lo, N, Cin, taps, stride, mode, stats, bias, gate, f32out=False):
    torch.manual_seed(B * 1000 + Lo + N + Cin + taps)
    code = ops.dtype_code(torch.empty(1, dtype=dt))
    Ls = Lo * stride if mode == 0 else (Lo // stride)
    W = (torch.randn(N, taps * Cin, device=dev) * 0.05).to(dt)
    A = torch.randn(B * Ls, Cin, device=dev).to(dt)
    bias_t = torch.randn(N, device=dev) if bias else None
    gate_t = torch.rand(B, N, device=dev) if gate else None
    outs = {}
    for deep in ("0", "5", "4"):
        os.environ["DRN_NT_DEEP"] = deep
        C = torch.full((B * Lo, N), 7.0, device=dev, dtype=torch.float32 if f32out else dt)
        st = torch.zeros((B * Lo + 127) // 128, 2, N, device=dev) if stats else None
        d = ops.gemm_desc(A, W, C, B * Lo, N, Cin, taps=taps, stride=stride, pad=(taps - 1) // 2, mode=mode, Lout=Lo, Lsrc=Ls,
                          stats=st, bias=bias_t, gate=gate_t, ldg=N, out_f32=f32out)
        ops.gemm_nt([d], code)
        torch.cuda.synchronize()
        outs[deep] = (C.float().clone(), None if st is None else st.clone())
    ok = True
    for deep in ("5", "4"):
        same = torch.equal(outs["0"][0], outs[deep][0])
        err = (outs["0"][0] - outs[deep][0]).abs().max().item()
        s_ok = True if not stats else torch.allclose(outs["0"][1], outs[deep][1], rtol=1e-5, atol=1e-4)
        print("  deep=%s identical=%s max|diff|=%.3e stats_ok=%s" % (deep, same, err, s_ok))
        ok = ok and same and s_ok
    return ok


cases = [
    # dt, B, Lo, N, Cin, taps, stride, mode, stats, bias, gate
    (torch.bfloat16, 32, 256, 4096, 4096, 1, 1, 0, False, True, True),
    (torch.bfloat16, 32, 256, 256, 4352, 3, 1, 0, True, False, False),
    (torch.bfloat16, 32, 256, 4352, 256, 3, 1, 1, False, False, False),
    (torch.bfloat16, 8, 128, 512, 256, 3, 2, 0, True, False, False),
    (torch.bfloat16, 8, 128, 256, 512, 3, 2, 1, False, False, False),
    (torch.bfloat16, 3, 100, 200, 64, 3, 1, 0, True, True, False),
    (torch.bfloat16, 5, 77, 136, 128, 1, 1, 0, False, True, True),
]
allok = True
for c in cases:
    print(c[1:])
    allok &= run(*c)
print("f32-output (weight gradient as NT)")
allok &= run(torch.bfloat16, 1, 4096, 4096, 8192, 1, 1, 0, False, False, False, f32out=True)
print("ALL OK" if allok else "MISMATCH")
