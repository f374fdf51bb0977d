"""Out-of-bounds WRITES of the streaming kernels (GPU): every output buffer is a window into a larger allocation filled
with a sentinel, at 16-byte-aligned and at unaligned offsets, for ragged element counts around the vector width, the
workgroup tile and the reduction chunk; after the launch the words on both sides of the window must still be the
sentinel.  (The bit-exact comparisons against the oracle check what is written INSIDE the window; a kernel that also wrote
past it would corrupt a neighbouring allocation of the caching allocator and go unnoticed there.)"""
import pytest
import torch

from torchdiffeq_amd.tableaus import DOPRI5, DOPRI8, carry_plan

pytestmark = pytest.mark.gpu
SENTINEL = 12345.0
PAD = 64
SIZES = [1, 3, 4, 5, 63, 255, 257, 1023, 1025, 2047, 2049, 4099, 65537]
DTYPES = [torch.float32, torch.float64]


class Window:
    """Output tensor carved out of a sentinel-filled allocation."""

    def __init__(self, n, dtype, offset):
        self.big = torch.full((PAD + offset + n + PAD,), SENTINEL, dtype=dtype, device="cuda")
        self.lo, self.hi = PAD + offset, PAD + offset + n
        self.t = self.big[self.lo:self.hi]

    def intact(self):
        return bool((self.big[:self.lo] == SENTINEL).all()) and bool((self.big[self.hi:] == SENTINEL).all())

    def written(self):
        return bool(torch.isfinite(self.t).all()) and not bool((self.t == SENTINEL).all())


def _rand(n, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, generator=g, dtype=torch.float64).to(dtype).cuda()


@pytest.mark.parametrize("offset", [0, 1, 3], ids=["aligned", "off1", "off3"])
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("n", SIZES)
def test_elementwise_kernels_write_only_their_window(hip_kernels, n, dtype, offset):
    k = hip_kernels
    y0, y1 = _rand(n, dtype, 1), _rand(n, dtype, 2)
    ks = [_rand(n, dtype, 10 + j) for j in range(14)]
    W = lambda: Window(n, dtype, offset)
    checks = []

    def run(name, wins, launch):
        launch()
        torch.cuda.synchronize()
        for i, w in enumerate(wins):
            checks.append((f"{name}[{i}]", w.intact(), w.written()))

    for tab in (DOPRI5, DOPRI8):
        for row in (tab.beta_rows()[0], tab.beta_rows()[-1]):
            w = W()
            run(f"stage_combine<{len(row.idx)}>", [w], lambda: k.stage_combine(w.t, y0, [ks[j] for j in row.idx], row.coef, 0.1))
    last = DOPRI5.beta_rows()[-1]
    w0, w1 = W(), W()
    run("stage_combine_err", [w0, w1], lambda: k.stage_combine_err(w0.t, w1.t, y0, [ks[j] for j in last.idx], last.coef,
                                                                   last.coef, 0.1))
    for name in ("dopri5", "dopri8"):
        plan = carry_plan(name)
        op = next(o for o in plan.ops if o is not None and len(o.targets) > 1 and not o.continues)
        wins = [W() for _ in op.targets]
        run(f"stage_combine_multi<{name}>", wins,
            lambda: k.stage_combine_multi([w.t for w in wins], op.spec, y0, None, [ks[j] for j in op.idx], 0.1))
    w = W()
    run("dense_eval", [w], lambda: k.dense_eval(w.t, y0, y1, ks[0], ks[6], ks[1:6], [0.1, -0.2, 0.3, 0.05, 0.4], 0.1, 0.37))
    m = 3
    rows = torch.full((PAD + offset + m * n + PAD,), SENTINEL, dtype=dtype, device="cuda")
    out_rows = rows[PAD + offset:PAD + offset + m * n].view(m, n)
    k.dense_eval_multi(out_rows, y0, y1, ks[0], ks[6], ks[1:6], [0.1, -0.2, 0.3, 0.05, 0.4], 0.1, [0.1, 0.5, 0.9])
    torch.cuda.synchronize()
    checks.append(("dense_eval_multi", bool((rows[:PAD + offset] == SENTINEL).all())
                   and bool((rows[PAD + offset + m * n:] == SENTINEL).all()), bool(torch.isfinite(out_rows).all())))
    fit = torch.full((PAD + offset + 5 * n + PAD,), SENTINEL, dtype=dtype, device="cuda")
    coeffs = fit[PAD + offset:PAD + offset + 5 * n].view(5, n)
    k.interp_fit(coeffs, y0, y1, ks[0], ks[6], ks[1:6], [0.1, -0.2, 0.3, 0.05, 0.4], 0.1)
    torch.cuda.synchronize()
    checks.append(("interp_fit", bool((fit[:PAD + offset] == SENTINEL).all())
                   and bool((fit[PAD + offset + 5 * n:] == SENTINEL).all()), bool(torch.isfinite(coeffs).all())))
    for stage, args in ((1, (ks[0], None, None, None)), (2, (ks[0], ks[1], None, None)), (3, (ks[0], ks[1], ks[2], None)),
                        (4, (ks[0], ks[1], ks[2], ks[3]))):
        w = W()
        run(f"rk4_stage{stage}", [w], lambda: k.rk4_stage(stage, w.t, y0, *args, 0.1))
    w = W()
    run("lerp", [w], lambda: k.lerp(w.t, y0, y1, 0.3))
    for mode in (0, 1):
        w = W()
        run(f"fixed_stage{mode}", [w], lambda: k.fixed_stage(mode, w.t, y0, ks[:2] if mode == 0 else ks[:1],
                                                             [0.25, 0.75] if mode == 0 else [0.5], 0.1))
    w = W()
    run("weighted_sum", [w], lambda: k.weighted_sum(w.t, ks[:8], [0.1 * (j + 1) for j in range(8)]))
    wins = [W() for _ in range(4)]
    run("scale_many", wins, lambda: k.scale_many([w.t for w in wins], y0, [0.5, -1.5, 2.0, 0.25]))
    wy, wd, wl = W(), W(), W()
    run("adams_predict", [wy, wd, wl], lambda: k.adams_predict(wy.t, y0, ks[:5], [0.1, -0.2, 0.3, -0.1, 0.05],
                                                               [0.2, 0.1, -0.3, 0.4, 0.1], 0.1, dy_out=wd.t, delta_out=wl.t))
    bad = [c for c in checks if not (c[1] and c[2])]
    assert not bad, bad


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("chunk", [1024, 2048])
def test_pack_segments_writes_only_its_chunks(hip_kernels, dtype, chunk):
    numels = [1, 5, chunk, 3 * chunk + 17, 700]
    starts, off = [], 0
    for m in numels:
        starts.append(off // chunk)
        off += -(-m // chunk) * chunk
    big = torch.full((PAD * 16 + off + PAD,), SENTINEL, dtype=dtype, device="cuda")
    out = big[PAD * 16:PAD * 16 + off]          # chunk data must stay 16-byte aligned: offset a multiple of 4 words
    srcs = [_rand(m, dtype, 30 + i) for i, m in enumerate(numels)]
    hip_kernels.pack_segments(out, srcs, starts, numels, [1.0, -1.0, 1.0, -1.0, 1.0], chunk)
    torch.cuda.synchronize()
    assert bool((big[:PAD * 16] == SENTINEL).all()) and bool((big[PAD * 16 + off:] == SENTINEL).all())
    assert bool(torch.isfinite(out).all())
