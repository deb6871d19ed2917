"""CPU emulation of split-precision GEMM arithmetic: what two fp16 pieces per operand and 3 products ("f16x3") would give
against the three bf16 pieces and 6 products in use ("bf16x6"), and bf16x3.  Products are formed exactly (fp64) from the
pieces, so only the representation / dropped-term error is measured (fp32 accumulation is common to all of them).
    python tools/split_numerics.py"""
import torch
torch.manual_seed(0)
M, K, N = 128, 2304, 512


def pieces(x, dtype, n):
    out, r = [], x.clone()
    for _ in range(n):
        p = r.to(dtype).to(torch.float32)
        out.append(p.double())
        r = r - p
    return out


def gemm(a, b, dtype, n, products):
    pa, pb = pieces(a, dtype, n), pieces(b, dtype, n)
    return sum(pa[i] @ pb[j] for i, j in products)


P6 = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
P3 = [(0, 0), (0, 1), (1, 0)]
for name, sa, sb in [('activations ~ N(0,1) x weights ~ N(0,1)/sqrt(K)', 1.0, K ** -0.5),
                     ('gradients ~ 1e-6 x weights', 1e-6, K ** -0.5),
                     ('gradients ~ 1e-6, scaled by 2^24 before the split', 1e-6 * 2 ** 24, K ** -0.5),
                     ('wide dynamic range: |x| log-uniform in [1e-6, 10]', None, K ** -0.5)]:
    if sa is None:
        a = torch.exp(torch.empty(M, K).uniform_(-13.8, 2.3)) * torch.sign(torch.randn(M, K))
    else:
        a = torch.randn(M, K) * sa
    b = torch.randn(K, N) * sb
    ref = a.double() @ b.double()
    scale = ref.abs().max()
    rows = []
    for label, dt, n, prod in [('bf16x6', torch.bfloat16, 3, P6), ('bf16x3', torch.bfloat16, 2, P3),
                               ('f16x3', torch.float16, 2, P3)]:
        err = ((gemm(a, b, dt, n, prod) - ref).abs().max() / scale).item()
        rows.append('%s %.1e' % (label, err))
    f32 = ((a @ b).double() - ref).abs().max() / scale
    print('%-55s fp32 GEMM %.1e | %s' % (name, f32.item(), ' | '.join(rows)))

# Small weights (the reference's conv_d initialisation: std sqrt(2 / (Cout * C * 3)) = 0.0128 at 64 channels, 0.0032 at 256):
# the residual plane of an unscaled weight is an fp16 subnormal; F16_W_SCALE = 2^8 (csrc/split_f16.h) restores it.
print()
K2 = 768
for std in (0.3, 0.0128, 0.0032, 0.001):
    a = torch.randn(M, K2) * 2.0 ** 12          # activations after their range scale
    b = torch.randn(K2, N) * std
    ref = a.double() @ b.double()
    scale = ref.abs().max()
    f32 = (((a @ b).double() - ref).abs().max() / scale).item()
    row = []
    for ws in (1.0, 256.0):
        r = gemm(a, b * ws, torch.float16, 2, P3) / ws
        row.append('weights x %-3g f16x3 %.1e' % (ws, ((r - ref).abs().max() / scale).item()))
    print('weights ~ N(0, %-6g)  fp32 GEMM %.1e | %s' % (std, f32, ' | '.join(row)))
