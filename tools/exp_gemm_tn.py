"""A/B of the outer-contiguous operand mapping (VAG_LIB=... selects the build): the model's weight-gradient shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
shapes = [("d out.weight cfg2", 9391, 256, 2560, 9392, 256), ("dec W_hh grad cfg2", 1536, 512, 2560, 1536, 512),
          ("dwp cfg2", 1536, 1024, 2560, 1536, 1024), ("dec grads cfg5", 3072, 1024, 20480, 3072, 1024),
          ("square", 4096, 4096, 4096, 4096, 4096)]
for name, M, N, K, lda, ldb in shapes:
    A = torch.randn(K, lda, device=dev); B = torch.randn(K, ldb, device=dev); C = torch.zeros(M, N, device=dev)
    def run():
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), 1, lda, L.ptr(B), ldb, 1, 1.0, L.ptr(C), N, None, 0, L.stream())
    run(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    print("%-20s %dx%dx%d TN beta=1 %9.1f us %6.1f TF/s" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
