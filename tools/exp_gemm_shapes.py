"""Experiment: TFLOP/s of vag_gemm_f32 on every large product of a cfg2 training step."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
R = 2560
# (name, M, N, K, layout, beta)  layout: NT = A(M,K) k-contig, B stored (N,K);  NN = B stored (K,N);  TN = A stored (K,M), B (K,N)
SHAPES = [("enc/dec in-proj", R, 1536, 256, "NT", 0), ("attn keys pe", R, 1024, 1024, "NT", 0),
          ("head W2", R, 256, 1024, "NT", 1), ("head W1", R, 256, 512, "NT", 0), ("logits", R, 9391, 256, "NT", 0),
          ("dW_out", 9391, 256, R, "TN", 1), ("d tmid", R, 256, 9391, "NN", 0), ("dW2", 256, 1024, R, "TN", 1),
          ("d_c head", R, 1024, 256, "NN", 0), ("d_h2 head", R, 512, 256, "NN", 0),
          ("dW_hh (3HxH)", 1536, 512, R, "TN", 1), ("dW_h (CxH)", 1024, 512, R, "TN", 1), ("dWp (3HxC)", 1536, 1024, R, "TN", 0),
          ("dW_ih2 chain", 1536, 512, 1024, "NT", 1), ("dW_c2h chain", 512, 1024, 1536, "TN", 1),
          ("dW_ih1 (3HxE)", 1536, 256, R, "TN", 1), ("de", R, 256, 1536, "NN", 0), ("d_enc pe", R, 1024, 1024, "NN", 1),
          ("dW_e", 1024, 1024, R, "TN", 1), ("Wp fold", 1536, 1024, 512, "NN", 0)]
tot_t = tot_f = 0.0
for name, M, N, K, lay, beta in SHAPES:
    if lay == "NT":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); sa = (K, 1); sb = (1, K)
    elif lay == "NN":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); sa = (K, 1); sb = (N, 1)
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); sa = (1, M); sb = (N, 1)
    ldc = (N + 3) // 4 * 4
    C = torch.zeros(M, ldc, device=dev)
    def run():
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(B), sb[0], sb[1], float(beta), L.ptr(C), ldc, None, 0, L.stream())
    run(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    reps = 20
    g = torch.cuda.CUDAGraph()           # graph replay: eager ctypes launches cost ~20 us of host time each
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay(); torch.cuda.synchronize()
    s.record()
    g.replay()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    fl = 2.0 * M * N * K
    tot_t += us; tot_f += fl
    print("%-16s %5dx%5dx%5d %s b%d  %8.1f us  %6.1f TF/s" % (name, M, N, K, lay, beta, us, fl / us / 1e6), flush=True)
print("sum %.1f us, %.1f GF -> %.1f TF/s" % (tot_t, tot_f / 1e9, tot_f / tot_t / 1e6))
