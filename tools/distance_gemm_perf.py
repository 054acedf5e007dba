"""MFMA distance-GEMM stress (BASELINE config 5 shape): bf16 rows d = 4096, query batch 4096.

    python tools/distance_gemm_perf.py [rows] [nq] [d]

Times isl_distance_matrix_bf16 (and the float32 GEMM for comparison) with HIP events through
torch and prints TFLOP/s against the dense MFMA peak of the dtype."""
import ctypes as C
import json
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch

import islands_amd as ia
from islands_amd import _ffi

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = torch.Generator(device=dev)
g.manual_seed(5)
rows = torch.nn.functional.normalize(torch.randn((n, d), device=dev, generator=g), dim=1)
q = torch.nn.functional.normalize(torch.randn((nq, d), device=dev, generator=g), dim=1)
out = torch.empty((nq, n), device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def sustained(fn, reps=20):
    """`reps` launches back to back inside one pair of events: the rate the chip sustains -- a single
    launch between two synchronisations starts on a chip that has dropped its clocks."""
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rb, qb = rows.to(torch.bfloat16).contiguous(), q.to(torch.bfloat16).contiguous()
bf16_call = lambda: ia._check(_ffi.lib().isl_distance_matrix_bf16(
    0, C.c_void_p(qb.data_ptr()), nq, C.c_void_p(rb.data_ptr()), n, d, C.c_void_p(out.data_ptr()), 1, 0, None))
ms = timed(bf16_call)
ms_s = sustained(bf16_call)
print(json.dumps({"op": "isl_distance_matrix_bf16 (cosine), 20 launches back to back", "ms": round(ms_s, 3),
                  "TFLOP/s": round(2.0 * nq * n * d / ms_s / 1e9, 1)}), flush=True)
ref = 1.0 - qb.float() @ rb.float().T
err = (out - ref).abs().max().item()
tf = 2.0 * nq * n * d / ms / 1e9
print(json.dumps({"op": "isl_distance_matrix_bf16 (cosine)", "nq": nq, "rows": n, "d": d, "ms": round(ms, 3),
                  "TFLOP/s": round(tf, 1), "frac_of_bf16_peak_2500": round(tf / 2500, 4),
                  "max_abs_diff_vs_torch_f32": err}), flush=True)
# the same call with the rows' and queries' sums of squares computed once (isl_row_sumsq_bf16) and handed
# in: what a caller with resident rows does (config 5: one pass over the 10M rows, reused by every batch)
rn = torch.empty(n, device=dev)
qn = torch.empty(nq, device=dev)
ia._check(_ffi.lib().isl_row_sumsq_bf16(C.c_void_p(rb.data_ptr()), n, d, C.c_void_p(rn.data_ptr()), 1, 0, None))
ia._check(_ffi.lib().isl_row_sumsq_bf16(C.c_void_p(qb.data_ptr()), nq, d, C.c_void_p(qn.data_ptr()), 1, 0, None))
out2 = torch.empty_like(out)
norms_call = lambda: ia._check(_ffi.lib().isl_distance_matrix_bf16_norms(
    0, C.c_void_p(qb.data_ptr()), nq, C.c_void_p(rb.data_ptr()), n, d, C.c_void_p(qn.data_ptr()), C.c_void_p(rn.data_ptr()),
    C.c_void_p(out2.data_ptr()), 1, 0, None))
ms_n = timed(norms_call)
ms_ns = sustained(norms_call)
same = bool((out2.view(torch.int32) == out.view(torch.int32)).all().item())
print(json.dumps({"op": "isl_distance_matrix_bf16_norms (cosine, sums of squares handed in)", "ms_isolated": round(ms_n, 3),
                  "TFLOP/s_isolated": round(2.0 * nq * n * d / ms_n / 1e9, 1), "ms_back_to_back": round(ms_ns, 3),
                  "TFLOP/s_back_to_back": round(2.0 * nq * n * d / ms_ns / 1e9, 1),
                  "bit_identical_to_isl_distance_matrix_bf16": same}), flush=True)
ms = timed(lambda: ia._check(_ffi.lib().isl_distance_matrix(
    0, C.c_void_p(q.data_ptr()), nq, C.c_void_p(rows.data_ptr()), n, d, C.c_void_p(out.data_ptr()), 1, 0, None)))
tf = 2.0 * nq * n * d / ms / 1e9
print(json.dumps({"op": "isl_distance_matrix f32 (cosine)", "nq": nq, "rows": n, "d": d, "ms": round(ms, 3),
                  "TFLOP/s": round(tf, 1), "frac_of_f32_peak_157": round(tf / 157.3, 4)}), flush=True)
