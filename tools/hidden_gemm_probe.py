import importlib, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
g = torch.Generator(device="cuda").manual_seed(0)
M, K = 201000, 3072
A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
for N in (128, 256):
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    for _ in range(2): eng.conv_gemm(A, W, N, K, T=201, bias=b, relu=True)
    eng.profile_begin()
    for _ in range(5): eng.conv_gemm(A, W, N, K, T=201, bias=b, relu=True)
    p = eng.profile_end()
    print(N, {k: round(v["ms"] / 5, 4) for k, v in p.items()}, flush=True)
