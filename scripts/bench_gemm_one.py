import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
M, N, K = 65536, 2048, 512
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, B.data_ptr(), K, None, 0, C.data_ptr(), N, st))
torch.cuda.synchronize()
