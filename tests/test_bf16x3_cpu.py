"""The arithmetic behind the fp32 path's "bf16x6" kernels (csrc/hip/bf16x3.h), restated in PyTorch on the CPU:
an fp32 value is the exact sum of three bf16 planes, and six bf16 x bf16 products accumulated in fp32 give a GEMM whose
error against float64 is that of an fp32 GEMM (the GPU tests hold the kernels themselves to the same oracle)."""
import torch


def split3(a):
    a0 = a.to(torch.bfloat16)
    r1 = a - a0.float()                 # exact in fp32: at most 16 significant bits left
    a1 = r1.to(torch.bfloat16)
    r2 = r1 - a1.float()                # exact again
    a2 = r2.to(torch.bfloat16)
    return a0, a1, a2, r2 - a2.float()


def test_three_bf16_planes_hold_an_fp32_value():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(1 << 16, generator=g) * torch.exp(torch.randn(1 << 16, generator=g) * 4)
    a0, a1, a2, rest = split3(a)
    s = a0.double() + a1.double() + a2.double()
    assert float(((s - a.double()).abs() / a.double().abs()).max()) <= 2.0 ** -25
    assert float((rest.abs() / a.abs()).max()) <= 2.0 ** -25
    # both residuals really were exact: adding the planes back in fp32 order reproduces the value bit for bit almost always
    assert float((((a0.float() + a1.float()) + a2.float()) == a).float().mean()) > 0.99


def test_six_products_match_an_fp32_gemm():
    g = torch.Generator().manual_seed(1)
    A = torch.relu(torch.randn(80, 800, generator=g))
    B = torch.randn(800, 64, generator=g) * 0.05
    ref = A.double() @ B.double()
    a = [p.float() for p in split3(A)[:3]]
    b = [p.float() for p in split3(B)[:3]]
    # (products of two bf16 values are exact in fp32; the matmuls below accumulate in fp32 like the MFMA)
    acc = torch.zeros(80, 64)
    for pa, pb in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
        acc = acc + a[pa] @ b[pb]
    err6 = float((acc.double() - ref).abs().max() / ref.abs().max())
    err32 = float(((A @ B).double() - ref).abs().max() / ref.abs().max())
    assert err6 < 1e-6 and err6 < 4 * err32 + 1e-7, (err6, err32)
    # what each dropped product would have cost
    five = acc - a[2] @ b[0]
    assert float((five.double() - ref).abs().max() / ref.abs().max()) > 3 * err6
