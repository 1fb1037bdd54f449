"""An EAGER two-stream probe for the exchange race of DESIGN.md section 3 ("Exchange protocol"): stream A launches a K-split product
over and over, two problems taking turns through the same workspace (a stale partial is then the other problem's and shows), every
result compared with its reference on the device; stream B runs an aggressor the whole time (AGGRESSOR=copy | cast | gemm).
RESULT (end of round 5): it does NOT reproduce the race -- the library from before the fix (BASE_LIB=1 DRN_LIB_PATH=...) ran
skinny_group_kernel 3 x 400 k launches beside each aggressor without one wrong tile, while the same kernel inside the replayed two-branch
hipGraph went wrong once in ~10^5 launches (scripts/experiments/forked_race_hunt.py).  So the zeros this probe reports for the GEMM
kernels' exchanges (200 k launches each, confirmation off and on) say nothing either way; the graph amplifier is the tool, and the GEMM
sites keep their confirmation wherever another queue can be beside them.  usage: python scripts/experiments/splitk_race_probe.py [launches]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from drn_amd import ops, _lib
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
BASE = os.environ.get("BASE_LIB") == "1"          # DRN_LIB_PATH points at a library from before the fix: the positive control
if BASE:
    ops.XCHG_CONFIRM = "2"                  # (no flag bits: what a library from before the flags understands)
g = torch.Generator().manual_seed(0)


def conv_case(B, L, Cin, Nout):
    M = B * L
    A = torch.randn(M, Cin, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(Nout, 3 * Cin, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    C = torch.empty(M, Nout, dtype=torch.bfloat16, device=dev)
    A2 = torch.randn(M, Cin, generator=g).to(torch.bfloat16).to(dev)
    return ([ops.gemm_desc(A, W, C, M, Nout, Cin, taps=3, pad=1, Lout=L, Lsrc=L)], [ops.gemm_desc(A2, W, C, M, Nout, Cin, taps=3, pad=1, Lout=L, Lsrc=L)]), C, (A, A2, W)


def plain_case(M, Nout, K):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(Nout, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    C = torch.empty(M, Nout, dtype=torch.bfloat16, device=dev)
    A2 = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    return ([ops.gemm_desc(A, W, C, M, Nout, K)], [ops.gemm_desc(A2, W, C, M, Nout, K)]), C, (A, A2, W)


def skinny_case():
    X = torch.randn(32, 4096, generator=g).to(dev)
    X2 = torch.randn(32, 4096, generator=g).to(dev)
    W = (torch.randn(1024, 4096, generator=g) * 0.02).to(dev)
    Y = torch.empty(32, 1024, device=dev)
    return (dict(X=X, W=W, Y=Y), dict(X=X2, W=W, Y=Y)), Y


cases = {"w4h conv0 8192x256x13056 (split 4)": conv_case(32, 256, 4352, 256),
         "general 2048x1024x1536 (split 2)": plain_case(2048, 1024, 1536),
         "general 512x512x4096": plain_case(512, 512, 4096)}
if os.environ.get("ONLY_SKINNY") == "1":
    cases = {}
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
AGG = os.environ.get("AGGRESSOR", "copy")            # copy: 1 GB device copies; cast: the step's throttled input cast; gemm: prop_fc's product
feat = torch.randn(8192, 4096, device=dev)
gA = torch.randn(8192, 4096, device=dev).to(torch.bfloat16)
gW = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
gC = torch.empty(8192, 4096, dtype=torch.bfloat16, device=dev)
gd = [ops.gemm_desc(gA, gW, gC, 8192, 4096, 4096)]


def aggress():
    if AGG == "copy":
        for _ in range(40):
            dst.copy_(src)                                   # ~0.45 ms each at 4.7 TB/s
    elif AGG == "cast":
        ops.CAST_THROTTLE = 192
        for _ in range(150):
            ops.cast_transpose(feat, ops.BF16)               # ~0.12 ms each, 192 resident workgroups (the two-branch step's setting)
        ops.CAST_THROTTLE = 0
    else:
        for _ in range(90):
            ops.gemm_nt(gd, ops.BF16)                        # ~0.2 ms each, one wave per SIMD on every CU
for name, (dd, C, keep) in cases.items():
    ks = ops._ksplit_w4h(dd[0], ops.BF16) if "w4h" in name else ops._ksplit(dd[0], ops.BF16)
    for conf in (("0", "1") if not BASE else ("base",)):
        for load in (True, False) if conf == "0" else (True,):
            ops.XCHG_CONFIRM = conf
            with torch.cuda.stream(sa):
                refs = []
                for descs in dd:                                 # two problems take turns through the same workspace: a stale
                    ops.gemm_nt(descs, ops.BF16)                 # partial is the OTHER problem's, so it shows
                    refs.append(C.clone())
                bad = torch.zeros((), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            t0 = time.time()
            done = 0
            while done < N:
                if load:
                    with torch.cuda.stream(sb):
                        aggress()
                with torch.cuda.stream(sa):
                    for i in range(500):
                        ops.gemm_nt(dd[i & 1], ops.BF16)
                        bad += (C != refs[i & 1]).any().long()
                done += 500
                torch.cuda.synchronize()
            print("%-34s ksplit %d confirm %s load %-5s: %d of %d launches differ  (%.1f s)" % (name, ks, conf, load, int(bad), done, time.time() - t0), flush=True)
# the kernel that did fail, as the positive control (its stores are always confirmed now: expect 0)
q, Y = skinny_case()
with torch.cuda.stream(sa):
    refs = []
    for qq in q:
        ops.skinny_group([qq])
        refs.append(Y.clone())
    bad = torch.zeros((), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
done = 0
while done < N:
    with torch.cuda.stream(sb):
        aggress()
    with torch.cuda.stream(sa):
        for i in range(500):
            ops.skinny_group([q[i & 1]])
            bad += (Y != refs[i & 1]).any().long()
    done += 500
    torch.cuda.synchronize()
print("skinny 32x1024x4096 (K-split, %s) beside %s" % ("UNCONFIRMED: the old library" if BASE else "confirmed", AGG) + ": %d of %d launches differ" % (int(bad), done), flush=True)
