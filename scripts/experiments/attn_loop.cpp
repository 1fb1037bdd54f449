// Per-launch GPU time of drn_qe_attn_fwd / drn_qe_attn_bwd inside a replayed hipGraph of 100 dependent launches (what the step
// is), next to an empty kernel's (the per-node floor of a linear graph).
// build+run (GPU box): hipcc --offload-arch=gfx950 -O3 scripts/experiments/attn_loop.cpp -Iinclude -Ldrn_amd -ldrn_hip -Wl,-rpath,$PWD/drn_amd -o /tmp/al && /tmp/al
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "drn_hip.h"
__global__ void empty_kernel() {}
int main() {
  const int B = 32, L = 8, C = 1024;
  float *out, *qcmd, *w, *bias, *att, *cmds, *dq, *dout, *dwp, *dbp; int64_t* lens;
  hipMalloc(&out, B * L * C * 4); hipMalloc(&qcmd, B * 3 * C * 4); hipMalloc(&w, C * 4); hipMalloc(&bias, 16); hipMalloc(&att, B * 3 * L * 4);
  hipMalloc(&cmds, 3 * B * C * 4); hipMalloc(&lens, B * 8); hipMalloc(&dq, B * 3 * C * 4); hipMalloc(&dout, B * L * C * 4);
  hipMalloc(&dwp, B * C * 4); hipMalloc(&dbp, B * 4);
  hipMemset(out, 0, B * L * C * 4); hipMemset(qcmd, 0, B * 3 * C * 4); hipMemset(w, 0, C * 4); hipMemset(bias, 0, 16);
  hipMemset(cmds, 0, 3 * B * C * 4);
  std::vector<int64_t> hl(B);
  for (int i = 0; i < B; ++i) hl[i] = 3 + i % 6;
  hipMemcpy(lens, hl.data(), B * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t st; hipStreamCreate(&st);
  for (int which = 0; which < 3; ++which) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; ++i) {
      if (which == 0) empty_kernel<<<1, 64, 0, st>>>();
      else if (which == 1) drn_qe_attn_fwd(out, qcmd, w, bias, lens, att, cmds, B, L, C, st);
      else drn_qe_attn_bwd(cmds, cmds + B * C, cmds + 2 * B * C, att, out, qcmd, w, lens, dq, dout, dwp, dbp, B, L, C, st);
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %.2f us per node (1000 nodes replayed)\n", which == 0 ? "empty" : (which == 1 ? "qe_attn_fwd" : "qe_attn_bwd"), ms);
  }
  return 0;
}
