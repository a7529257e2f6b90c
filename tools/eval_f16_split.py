"""CPU emulation of candidate product formations for resel_gemm_f32: error against an fp64 product,
relative to sum|a b|, on the operands of tests/test_hip_ops.py::test_gemm_f32_split_modes_error_against_fp64.
  f32     the fp32 MFMA (exact products, one fp32 rounding per accumulated term: an fmaf chain)
  bf16x6  mode 6: three truncated bf16 planes, six leading plane products
  f16x3   two fp16 planes (round to nearest), per-TENSOR power-of-two scale, residual plane scaled by 2^11 into its own
          accumulator, three plane products
Matrix-core accumulation is modelled two ways: 'blk' = the 16 products of one instruction summed exactly, one rounding per
instruction; 'seq' = one rounding per product (pessimistic)."""
import numpy as np


def planes_bf16(x, n):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        p = (r.view(np.uint32) & 0xffff0000).view(np.float32)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def planes_f16(x):
    s = 2.0 ** (15 - np.ceil(np.log2(np.abs(x).max())))            # max |x s| in [2^14, 2^15]
    xs = (x * np.float32(s)).astype(np.float32)
    h1 = xs.astype(np.float16).astype(np.float32)
    h2 = ((xs - h1) * np.float32(2048.0)).astype(np.float16).astype(np.float32)
    return h1, h2, s


def acc(terms, mode):
    """terms: list of (A_plane [M,K], B_plane [N,K]) accumulated into one fp32 accumulator, k blocks of 16."""
    M, K = terms[0][0].shape
    N = terms[0][1].shape[0]
    c = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        for a, b in terms:
            if mode == 'blk':
                c = (c.astype(np.float64) + a[:, k0:k0 + 16].astype(np.float64) @ b[:, k0:k0 + 16].astype(np.float64).T).astype(np.float32)
            else:
                for k in range(k0, min(K, k0 + 16)):
                    c = (c.astype(np.float64) + np.outer(a[:, k].astype(np.float64), b[:, k].astype(np.float64))).astype(np.float32)
    return c


def main():
    rs = np.random.RandomState(5)
    M, N, K = 48, 40, 4096
    for name, gen in (('wide (6 decades)', lambda *s: (rs.randn(*s) * np.exp(rs.randn(*s) * 2.0)).astype(np.float32)),
                      ('gaussian', lambda *s: rs.randn(*s).astype(np.float32))):
        A, B = gen(M, K), gen(N, K)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
        print(f'--- {name}, K = {K}: error / sum|ab|  (mean, max)')
        for mode in ('blk', 'seq'):
            res = {}
            res['f32'] = acc([(A, B)], 'seq')               # the fp32 instruction has K = 2 per issue: one rounding per term either way
            a, b = planes_bf16(A, 3), planes_bf16(B, 3)
            res['bf16x6'] = acc([(a[2], b[0]), (a[0], b[2]), (a[1], b[1]), (a[1], b[0]), (a[0], b[1]), (a[0], b[0])], mode)
            a1, a2, sa = planes_f16(A)
            b1, b2, sb = planes_f16(B)
            main_ = acc([(a1, b1)], mode)
            cross = acc([(a1, b2), (a2, b1)], mode)
            res['f16x3'] = ((main_.astype(np.float64) + cross.astype(np.float64) / 2048.0) / (sa * sb)).astype(np.float32)
            for k, v in res.items():
                e = np.abs(v.astype(np.float64) - ref) / scale
                print(f'  [{mode}] {k:7s} mean {e.mean():.3e}  max {e.max():.3e}   x f32: mean {e.mean() / (np.abs(res["f32"].astype(np.float64) - ref) / scale).mean():.2f} '
                      f'max {e.max() / (np.abs(res["f32"].astype(np.float64) - ref) / scale).max():.2f}')


if __name__ == '__main__':
    main()
