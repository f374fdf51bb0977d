"""Butcher tableaus of the in-scope explicit Runge–Kutta methods, as host-side coefficient tables.

Published constants (Dormand & Prince 1980 / Shampine 1986 for the 5(4) pair and its midpoint
weights; Prince & Dormand 1981 for the 8(7) 13-stage pair).  The reference holds the same numbers as
`torch.float64` tensors (torchdiffeq/_impl/dopri5.py:5-30, dopri8.py:5-70); here they are kept as
rational text and turned into doubles the same way the reference's Python literals are evaluated
(`num / den` in double; error weights as the double difference `b - b_hat`), so both libraries feed
bit-identical fp64 coefficients to the kernels (checked in tests/test_tableaus.py against
tests/golden/tableaus.npz).

The kernels skip structural zeros, so every row is stored sparse: `(stage indices, coefficients)`.
"""
from __future__ import annotations

import dataclasses
import functools
import math
from typing import List, Sequence, Tuple

import numpy as np


def _ratio(tok: str) -> float:
    """'a/b' -> float(a)/float(b) (one rounding, like the Python literal `a / b`)."""
    tok = tok.strip()
    if "/" in tok:
        num, den = tok.split("/")
        return float(int(num)) / float(int(den))
    return float(tok)


def _row(text: str) -> List[float]:
    return [_ratio(tok) for tok in text.split()]


class RowCoefs(tuple):
    """The non-zero weights of a tableau row — a plain tuple to everything that only multiplies (the HIP kernels take the
    values; equality and hashing are the tuple's) — that also remembers WHERE in the dense row they sit (`idx`) and how
    long that row is (`width`).  The torch-op host path needs both: ATen's `torch.sum` over the stage dimension adds the
    products in an order that depends on their positions (four interleaved partial sums, vector lanes from 8 columns on;
    docs/LAB_NOTEBOOK.md §8), and reproducing the reference bit for bit there means handing ATen the same dense row."""

    def __new__(cls, values, idx, width):
        self = super().__new__(cls, values)
        self.idx, self.width = tuple(idx), int(width)
        return self


@dataclasses.dataclass(frozen=True)
class SparseRow:
    idx: Tuple[int, ...]      # stage slots with a non-zero weight
    coef: "RowCoefs"          # fp64 weights (cast to the state dtype by the kernel), a tuple that knows its row

    @staticmethod
    def from_dense(values: Sequence[float]) -> "SparseRow":
        return _sparse_row(tuple(float(v) for v in values))

    @staticmethod
    def literal(values: Sequence[float]) -> "SparseRow":
        """EVERY slot of the dense row, the zero weights too — the row as the reference multiplies it
        (`k * (beta_i * dt)`, rk_common.py:79-89: `inf * 0` is NaN).  For the torch-op host path, which evaluates the
        reference's expressions literally; the kernels never read a zero-weight stage (docs/LAB_NOTEBOOK.md §8)."""
        return _literal_row(tuple(float(v) for v in values))


@functools.lru_cache(maxsize=None)
def _sparse_row(values: Tuple[float, ...]) -> SparseRow:
    """Memoised: the rows of the (few, immutable) tableaus are rebuilt by every solver construction otherwise — the
    adjoint constructs one solver per output interval."""
    nz = [(i, v) for i, v in enumerate(values) if v != 0.0]
    if not nz and values:
        # an all-zero row of a user-defined table (a stage that restarts from y0, an error row that estimates nothing):
        # the kernels take >= 1 term, so keep one explicit zero — 0 * k_0, which is also what the reference's dense
        # sum computes (rk_common.py:79, :88), non-finite k_0 included
        nz = [(0, 0.0)]
    idx = tuple(i for i, _ in nz)
    return SparseRow(idx, RowCoefs((v for _, v in nz), idx, len(values)))


@functools.lru_cache(maxsize=None)
def _literal_row(values: Tuple[float, ...]) -> SparseRow:
    idx = tuple(range(len(values)))
    return SparseRow(idx, RowCoefs(values, idx, len(values)))


@dataclasses.dataclass(frozen=True)
class Tableau:
    name: str
    order: int
    alpha: Tuple[float, ...]          # stage abscissae c_i, i = 1..S
    beta: Tuple[Tuple[float, ...], ...]   # dense lower-triangular rows a_ij
    c_sol: Tuple[float, ...]          # solution weights b_j over the S+1 stage slots
    c_error: Tuple[float, ...]        # b_j - b_hat_j over the S+1 stage slots
    c_mid: Tuple[float, ...]          # weights of y(t0 + dt/2) (dense output), S+1 slots

    @property
    def n_stages(self) -> int:
        return len(self.alpha)

    @property
    def fsal_solution(self) -> bool:
        """True when the last stage input already equals y1 (rk_common.py:83: no extra combine)."""
        return self.c_sol[-1] == 0.0 and tuple(self.c_sol[:-1]) == tuple(self.beta[-1])

    def beta_rows(self, literal: bool = False) -> List[SparseRow]:
        return [(SparseRow.literal if literal else SparseRow.from_dense)(r) for r in self.beta]

    def dense(self):
        """Dense fp64 numpy views (alpha, beta rows, c_sol, c_error, c_mid) for tests."""
        return (np.array(self.alpha), [np.array(r) for r in self.beta], np.array(self.c_sol),
                np.array(self.c_error), np.array(self.c_mid))


# ---------------------------------------------------------------------------------------------------
# Dormand–Prince 5(4), 7 stage slots (FSAL).  dopri5.py:5-30
# ---------------------------------------------------------------------------------------------------
_DP5_ALPHA = "1/5 3/10 4/5 8/9 1 1"
_DP5_A = """
1/5
3/40 9/40
44/45 -56/15 32/9
19372/6561 -25360/2187 64448/6561 -212/729
9017/3168 -355/33 46732/5247 49/176 -5103/18656
35/384 0 500/1113 125/192 -2187/6784 11/84
"""
_DP5_B = "35/384 0 500/1113 125/192 -2187/6784 11/84 0"
_DP5_BHAT = "1951/21600 0 22642/50085 451/720 -12231/42400 649/6300 1/60"
# Shampine's midpoint weights, stored as 2*w (the table value is halved below).
_DP5_MID2 = ("6025192743/30085553152 0 51252292925/65400821598 -2691868925/45128329728 "
             "187940372067/1594534317056 -1776094331/19743644256 11237099/235043384")


def _dopri5() -> Tableau:
    b = _row(_DP5_B)
    bhat = _row(_DP5_BHAT)
    err = [bj - bh for bj, bh in zip(b, bhat)]
    err[-1] = -1.0 / 60.0
    mid = [v / 2 for v in _row(_DP5_MID2)]
    rows = tuple(tuple(_row(line)) for line in _DP5_A.strip().splitlines())
    return Tableau("dopri5", 5, tuple(_row(_DP5_ALPHA)), rows, tuple(b), tuple(err), tuple(mid))


# ---------------------------------------------------------------------------------------------------
# Prince–Dormand 8(7), 14 stage slots (FSAL).  dopri8.py:5-70
# ---------------------------------------------------------------------------------------------------
_DP8_ALPHA = ("1/18 1/12 1/8 5/16 3/8 59/400 93/200 5490023248/9719169821 13/20 "
              "1201146811/1299019798 1 1 1")
_DP8_A = """
1/18
1/48 1/16
1/32 0 3/32
5/16 0 -75/64 75/64
3/80 0 0 3/16 3/20
29443841/614563906 0 0 77736538/692538347 -28693883/1125000000 23124283/1800000000
16016141/946692911 0 0 61564180/158732637 22789713/633445777 545815736/2771057229 -180193667/1043307555
39632708/573591083 0 0 -433636366/683701615 -421739975/2616292301 100302831/723423059 790204164/839813087 800635310/3783071287
246121993/1340847787 0 0 -37695042795/15268766246 -309121744/1061227803 -12992083/490766935 6005943493/2108947869 393006217/1396673457 123872331/1001029789
-1028468189/846180014 0 0 8478235783/508512852 1311729495/1432422823 -10304129995/1701304382 -48777925059/3047939560 15336726248/1032824649 -45442868181/3398467696 3065993473/597172653
185892177/718116043 0 0 -3185094517/667107341 -477755414/1098053517 -703635378/230739211 5731566787/1027545527 5232866602/850066563 -4093664535/808688257 3962137247/1805957418 65686358/487910083
403863854/491063109 0 0 -5068492393/434740067 -411421997/543043805 652783627/914296604 11173962825/925320556 -13158990841/6184727034 3936647629/1978049680 -160528059/685178525 248638103/1413531060 0
14005451/335480064 0 0 0 0 -59238493/1068277825 181606767/758867731 561292985/797845732 -1041891430/1371343529 760417239/1151165299 118820643/751138087 -528747749/2220607170 1/4
"""
_DP8_B = ("14005451/335480064 0 0 0 0 -59238493/1068277825 181606767/758867731 561292985/797845732 "
          "-1041891430/1371343529 760417239/1151165299 118820643/751138087 -528747749/2220607170 1/4 0")
_DP8_BHAT = ("13451932/455176623 0 0 0 0 -808719846/976000145 1757004468/5645159321 "
             "656045339/265891186 -3867574721/1518517206 465885868/322736535 53011238/667516719 2/45 0 0")
# Dense-output polynomials b_j(theta) (descending powers theta^5..theta^1, then the constant term),
# evaluated at theta = 1/2 for the midpoint weights.  slot -> coefficients.
_DP8_DENSE = {
    0: "-6.3448349392860401388 22.1396504998094068976 -30.0610568289666450593 19.9990069333683970610 -6.6910181737837595697 1.0",
    5: "-39.6107919852202505218 116.4422149550342161651 -121.4999627731334642623 52.2273532792945524050 -7.6142658045872677172",
    6: "20.3761213808791436958 -67.1451318825957197185 83.1721004639847717481 -46.8919164181093621583 10.7281392630428866124",
    7: "7.3347098826795362023 -16.5672243527496524646 9.5724507555993664382 -0.1890893225010595467 0.5526637063753648783",
    8: "32.8801774352459155182 -89.9916014847245016028 87.8406057677205645007 -35.7075975946222072821 4.2186562625665153803",
    9: "-10.1588990526426760954 22.6237489648532849093 -17.4152107770762969005 6.2736448083240352160 -0.6627209125361597559",
    10: "-12.5401268098782561200 32.2362340167355370113 -28.5903289514790976966 10.3160881272450748458 -1.2636789001135462218",
    11: "29.5553001484516038033 -82.1020315488359848644 81.6630950584341412934 -34.7650769866611817349 5.4106037898590422230",
    12: "-41.7923486424390588923 116.2662185791119533462 -114.9375291377009418170 47.7457971078225540396 -7.0321379067945741781",
    13: "20.3006925822100825485 -53.9020777466385396792 50.2558364226176017553 -19.0082099341608028453 2.3537586759714983486",
}


def _dense_weight(coeffs: Sequence[float], theta: float) -> float:
    """sum_p c_p theta^p (left to right, descending powers) scaled by theta — dopri8.py:31-66."""
    total = None
    for power, c in zip((5, 4, 3, 2, 1), coeffs[:5]):
        term = c * (theta ** power)
        total = term if total is None else total + term
    if len(coeffs) > 5:
        total = total + coeffs[5]
    return total / (1 / theta)


def _dopri8() -> Tableau:
    b = _row(_DP8_B)
    bhat = _row(_DP8_BHAT)
    err = [bj - bh for bj, bh in zip(b, bhat)]
    err[12], err[13] = 1.0 / 4.0, 0.0
    mid = [0.0] * 14
    for slot, text in _DP8_DENSE.items():
        mid[slot] = _dense_weight([float(tok) for tok in text.split()], 0.5)
    rows = tuple(tuple(_row(line)) for line in _DP8_A.strip().splitlines())
    return Tableau("dopri8", 8, tuple(_row(_DP8_ALPHA)), rows, tuple(b), tuple(err), tuple(mid))


DOPRI5 = _dopri5()
DOPRI8 = _dopri8()


# ---------------------------------------------------------------------------------------------------
# Tsitouras 5(4) (Tsitouras 2011, "Runge–Kutta pairs of order 5(4) satisfying only the first column
# simplifying assumption"), 7 stage slots.  tsit5.py:6-73.  Published constants, 20 significant digits
# (they round to the same doubles as the reference's longer literals — pinned by tests/test_tableaus.py).
# The reference's table ends `c_sol` with 1/66 in slot 6, so rk_common.py:83's FSAL shortcut does NOT
# apply: y1 is an extra 7-term combine and f1 = k_6 is the derivative at the last stage input.
# ---------------------------------------------------------------------------------------------------
_TS5_ALPHA = "0.161 0.327 0.9 0.98002554090450968573 1 1"
_TS5_A = """
0.161
-0.0084806554923569885444 0.33548065549235698854
2.8971530571054934321 -6.3594484899750748431 4.3622954328695814110
5.3258648284392566044 -11.748883564062827878 7.4955393428898362083 -0.092495066361755249257
5.8614554429464200287 -12.920969317847109292 8.1593678985761586432 -0.071584973281400997225 -0.028269050394068382909
0.096460766818065229518 0.01 0.47988965041449957478 1.3790085741037418932 -3.2900695154360806799 2.3247105240997739824
"""
_TS5_B = ("0.094680755765839458075 0.0091835655403432530968 0.48777052842476157079 1.2342975669304789857 "
          "-2.7077123499835254549 1.8666284181705870358 1/66")
_TS5_ERR = ("-0.0017800110522257714434 -0.00081643445965674690322 0.0078808780102619960103 "
            "-0.14471100717326290754 0.58235716545255522502 -0.45808210592918694666 1/66")


def _tsit5_dense(x: float) -> List[float]:
    """Tsitouras' continuous-extension weights b_j(x) in their published factored form (tsit5.py:66-76;
    the second factor of b_1 uses the reference's root 1.329989018975412)."""
    return [
        -1.0530884977290216 * x * (x - 1.329989018975412) * (x * x - 1.4364028541716351 * x + 0.7139816917074209),
        0.1017 * x * x * (x * x - 2.1966568338249754 * x + 1.2949852507374631),
        2.490627285651252793 * x * x * (x * x - 2.38535645472061657 * x + 1.57803468208092486),
        -16.54810288924490272 * (x - 1.21712927295533244) * (x - 0.61620406037800089) * x * x,
        47.37952196281928122 * (x - 1.203071208372362603) * (x - 0.658047292653547382) * x * x,
        -34.87065786149660974 * (x - 1.2) * (x - 2 / 3) * x * x,
        2.5 * (x - 1) * (x - 0.6) * x * x,
    ]


def _tsit5() -> Tableau:
    rows = tuple(tuple(_row(line)) for line in _TS5_A.strip().splitlines())
    return Tableau("tsit5", 5, tuple(_row(_TS5_ALPHA)), rows, tuple(_row(_TS5_B)), tuple(_row(_TS5_ERR)),
                   tuple(_tsit5_dense(1 / 2)))


# ---------------------------------------------------------------------------------------------------
# Bogacki–Shampine 3(2), 4 stage slots (FSAL).  bosh3.py:5-17
# ---------------------------------------------------------------------------------------------------
def _bosh3() -> Tableau:
    b = _row("2/9 1/3 4/9 0")
    bhat = _row("7/24 1/4 1/3 1/8")
    err = [bj - bh for bj, bh in zip(b, bhat)]
    rows = (tuple(_row("1/2")), tuple(_row("0 3/4")), tuple(_row("2/9 1/3 4/9")))
    return Tableau("bosh3", 3, tuple(_row("1/2 3/4 1")), rows, tuple(b), tuple(err), (0.0, 0.5, 0.0, 0.0))


# ---------------------------------------------------------------------------------------------------
# Fehlberg 2(1), 3 stage slots (solution weights differ from the last row: extra combine).  fehlberg2.py:4-17
# ---------------------------------------------------------------------------------------------------
def _fehlberg2() -> Tableau:
    b = _row("1/512 255/256 1/512")
    bhat = _row("1/256 255/256 0")
    err = [bj - bh for bj, bh in zip(b, bhat)]
    rows = (tuple(_row("1/2")), tuple(_row("1/256 255/256")))
    return Tableau("fehlberg2", 2, tuple(_row("1/2 1")), rows, tuple(b), tuple(err), (0.0, 0.5, 0.0))


# ---------------------------------------------------------------------------------------------------
# Heun–Euler 2(1) ("adaptive_heun"), 2 stage slots.  adaptive_heun.py:5-20
# ---------------------------------------------------------------------------------------------------
def _adaptive_heun() -> Tableau:
    b = (0.5, 0.5)
    bhat = (0.0, 1.0)
    err = tuple(bj - bh for bj, bh in zip(b, bhat))
    return Tableau("adaptive_heun", 2, (1.0,), ((1.0,),), b, err, (0.5, 0.0))


TSIT5 = _tsit5()
BOSH3 = _bosh3()
FEHLBERG2 = _fehlberg2()
ADAPTIVE_HEUN = _adaptive_heun()

ADAPTIVE_TABLEAUS = {t.name: t for t in (DOPRI8, DOPRI5, TSIT5, BOSH3, FEHLBERG2, ADAPTIVE_HEUN)}


# ---------------------------------------------------------------------------------------------------
# Carried partial sums (r03; include/tdeq_hip.h `tdeq_stage_combine_multi`).
#
# Row i of `_runge_kutta_step` (rk_common.py:69-81) is y_i = y0 + sum_{j<=i} (beta_ij dt) k_j: launched row by row,
# every row re-reads all earlier stages — 32 words per element and dopri5 step, 94 per dopri8 step.
# While a "host" row h is being formed its stages are in registers, so the same launch can also emit, for a later
# "target" row t, the left-to-right prefix of t's sum over the stages <= h; row t then reads that one stream plus the
# stages computed since (and y0).  The embedded error (rk_common.py:89) is one more target whose last consumer is the
# norm kernel.  A target whose row has no weight on any stage after h needs no later launch at all: the host writes
# the finished stage input (dopri8's row for stage 12 has a zero weight on k_11, dopri8.py:5-70).
# The left-to-right order of every sum is unchanged, so all stage inputs are bit-identical to the row-by-row launches.
#
# The assignment target -> host below is the minimum of the word count over all assignments with at most
# TDEQ_MAX_MULTI_OUT outputs per launch, hosts that are themselves read in full, and at most two error terms left to
# the norm kernel (found by annealing over the assignment space, tools/carry_search.py; `CarryPlan.words` recomputes
# the count, tests/test_carry.py pins it):  dopri5 37 -> 35 words per element and step, dopri8 98 -> 75 (13 launches
# instead of 14).  Target S (= number of rows) is the embedded error.
# ---------------------------------------------------------------------------------------------------
_CARRY_HOSTS = {
    "dopri5": {4: 3, 6: 5},
    "dopri8": {5: 4, 7: 6, 8: 6, 9: 6, 11: 10, 12: 10, 13: 10},
    # tsit5's solution is not its last stage input (c_sol ends in 1/66): launch row 6 is the c_sol combine, target 7 the
    # error.  Row 2 hosts row 3; row 4 hosts row 5, the solution and the error: 46 -> 41 words.
    "tsit5": {3: 2, 5: 4, 6: 4, 7: 4},
}
MAX_MULTI_OUT = 4
# Where the plan is switched on without being asked for (TDEQ_CARRY unset): tableau -> smallest state (elements) it
# applies to.  Measured on the MI355X (profiles/r03_carry_bench.json, r03_cfg2_carry1_kernel_stats.csv) — dopri8:
# solver kernels -25 % (fp64 16384x512) / -22 % (fp32 65536x128), trial step -12 % / -16 %, also on the 1/8 shards (one
# launch fewer); dopri5: 2 of 37 words — rows 4 + 5 take 33.8 + 21.5 us instead of 29.6 + 36.1 at 65536x128 (trial step
# -1.5 %), but the two-output launch costs more than it saves where a launch is latency, not bytes (1/8 shard: +0.8 %),
# so dopri5 takes the plan from 4 M elements on; tsit5 likewise (trial step -3.6 % fp32 at 65536x128, -6.9 % fp64 at
# 16384x512, profiles/r03_carry_bench_tsit5.json).
CARRY_DEFAULT_ON = {"dopri8": 0, "dopri5": 1 << 22, "tsit5": 1 << 22}


@dataclasses.dataclass(frozen=True)
class CarryOp:
    """One launch of the planned stage loop: forms the stage input of `row` (output 0) and the carried outputs."""
    row: int
    idx: Tuple[int, ...]                    # stage slots read, ascending
    continues: bool                         # output 0 continues the partial sum carried for `row`
    targets: Tuple[int, ...]                # tableau row of every output (targets[0] == row; n_rows = the error)
    spec: Tuple[Tuple[Tuple[float, ...], int, bool], ...]   # per output: (weights over idx, mask, add_y0)


@dataclasses.dataclass(frozen=True)
class CarryPlan:
    ops: Tuple[object, ...]                 # per tableau row: a CarryOp, or None (input finished by an earlier launch;
                                            # row 0 keeps its own launch forms and is None as well)
    err_idx: Tuple[int, ...]                # stages the norm kernel adds to the carried partial error
    err_coef: Tuple[float, ...]
    words: int                              # words per element and step moved by the planned launches + the norm
    launches: int


def _plan_from_hosts(tab: Tableau, hosts) -> CarryPlan:
    """Launch rows 0..S-1 form the stage inputs; a pair whose solution is not its last stage input (tsit5) has one more
    launch row S, the solution combine over c_sol (no evaluation follows it).  The embedded error is target R = the
    number of launch rows."""
    rows = tab.beta_rows()
    S = len(rows)
    R = S if tab.fsal_solution else S + 1
    err = SparseRow.from_dense(tab.c_error)
    nz = {i: dict(zip(r.idx, r.coef)) for i, r in enumerate(rows)}
    if not tab.fsal_solution:
        sol = SparseRow.from_dense(tab.c_sol)
        nz[S] = dict(zip(sol.idx, sol.coef))
    nz[R] = dict(zip(err.idx, err.coef))
    assert hosts.get(R) is not None
    for t, h in hosts.items():
        assert 1 <= h < t <= R and h not in hosts, "hosts are rows that are formed in full"
        assert any(j <= h for j in nz[t]), "nothing to carry"
    ops, finished, words, launches = [None] * R, set(), 0, 0
    for i in range(R):
        if i in finished:
            continue
        h = hosts.get(i)
        own = sorted(j for j in nz[i] if h is None or j > h)
        if h is not None and not own:
            raise AssertionError("a row without newer stages is finished by its host")
        targets = [i] + sorted(t for t, hh in hosts.items() if hh == i)
        assert len(targets) <= MAX_MULTI_OUT
        idx = sorted(set(own) | {j for t in targets[1:] for j in nz[t] if j <= i})
        spec = []
        for t in targets:
            take = own if t == i else [j for j in sorted(nz[t]) if j <= i]
            mask = sum(1 << idx.index(j) for j in take)
            coefs = tuple(nz[t].get(j, 0.0) if j in take else 0.0 for j in idx)
            done = t == i or (t < S and all(j <= i for j in nz[t]))
            if done and t != i:
                finished.add(t)
            spec.append((coefs, mask, done))
        words += len(idx) + 1 + (1 if h is not None else 0) + len(targets)
        launches += 1
        if i > 0:
            ops[i] = CarryOp(i, tuple(idx), h is not None, tuple(targets), tuple(spec))
        else:
            assert targets == [0]
    rem = sorted(j for j in nz[R] if j > hosts[R])
    assert len(rem) <= 2, "tdeq_error_norm_partial continues over at most two stages"
    words += 1 + len(rem) + 2
    launches += 1
    return CarryPlan(tuple(ops), tuple(rem), tuple(nz[R][j] for j in rem), words, launches)


@functools.lru_cache(maxsize=None)
def carry_plan(name: str):
    """The carried-partial-sum launch plan of the tableau `name`, or None when it has none (no saving: bosh3,
    fehlberg2, adaptive_heun)."""
    hosts = _CARRY_HOSTS.get(name)
    return None if hosts is None else _plan_from_hosts(ADAPTIVE_TABLEAUS[name], dict(hosts))


def row_by_row_words(tab: Tableau) -> int:
    """Words per element and step of the row-by-row launches with the end-of-step fusion (docs/LAB_NOTEBOOK.md §3)."""
    rows = tab.beta_rows()
    err = SparseRow.from_dense(tab.c_error)
    last = rows[-1] if tab.fsal_solution else SparseRow.from_dense(tab.c_sol)
    return sum(len(r.idx) + 2 for r in rows) + (0 if tab.fsal_solution else len(last.idx) + 2) + 1 + \
        (1 + len(err.idx) - len(last.idx) + 2)


# ---------------------------------------------------------------------------------------------------
# Adams–Bashforth / Adams–Moulton weights (fixed_adams.py:10-152 holds them as integer tables over a common
# divisor).  They are the integrals over one step of the Lagrange basis polynomials through the last k derivative
# values (Bashforth: nodes t_n, t_{n-1}, ...; Moulton: t_{n+1}, t_n, ...), generated here in exact rational
# arithmetic and rounded once — the same doubles as the reference's `numerator / divisor` (Python's int / int is
# correctly rounded), checked against the reference's tables in tests/test_tableaus.py (golden/adams.npz).
# ---------------------------------------------------------------------------------------------------
def _lagrange_step_integrals(k: int, shift: int) -> Tuple[float, ...]:
    from fractions import Fraction
    out = []
    for j in range(k):
        poly = [Fraction(1)]                    # prod_{i != j} (u - x_i), x_i = shift - i, ascending powers of u
        den = Fraction(1)
        for i in range(k):
            if i == j:
                continue
            root = Fraction(shift - i)
            nxt = [Fraction(0)] * (len(poly) + 1)
            for p, c in enumerate(poly):
                nxt[p] -= root * c
                nxt[p + 1] += c
            poly = nxt
            den *= Fraction(i - j)               # x_j - x_i
        integral = sum(c / (p + 1) for p, c in enumerate(poly))      # over u in [0, 1]
        out.append(float(integral / den))
    return tuple(out)


@functools.lru_cache(maxsize=None)
def adams_coefficients(order: int) -> Tuple[Tuple[float, ...], Tuple[float, ...]]:
    """(Bashforth weights, Moulton weights) of the given order, newest derivative first; 1 <= order <= 20."""
    if not 1 <= order <= 20:
        raise ValueError("Adams coefficients are available for orders 1..20")
    return _lagrange_step_integrals(order, 0), _lagrange_step_integrals(order, 1)


# ---------------------------------------------------------------------------------------------------
# Implicit Runge–Kutta tableaus of the fixed-grid implicit solvers (fixed_grid_implicit.py:10-140): published
# collocation / DIRK coefficients (Gauss–Legendre, Radau IIA, Alexander's SDIRK2, TR-BDF2), written as the
# closed forms in surds and evaluated in double — checked bit for bit against the reference's tensors in
# tests/test_implicit_golden.py (golden/implicit.npz).
# ---------------------------------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class ImplicitTableau:
    name: str
    order: int
    alpha: Tuple[float, ...]
    beta: Tuple[Tuple[float, ...], ...]   # full rows (FIRK) or lower-triangular rows incl. the diagonal (DIRK)
    c_sol: Tuple[float, ...]
    diagonal: bool                         # DIRK: stages solved one after the other


def _implicit_tableaus():
    # The reference takes its surds from torch.sqrt on 0-dim fp64 tensors (fixed_grid_implicit.py:5-8); for 2 that
    # returns 1.414213562373095, one ulp below the correctly rounded math.sqrt(2.0) — kept, so that sdirk2 / trbdf2
    # carry the reference's coefficients bit for bit.
    r2, r3, r6, r15 = 1.414213562373095, math.sqrt(3.0), math.sqrt(6.0), math.sqrt(15.0)
    tabs = [
        ImplicitTableau("implicit_euler", 1, (1.0,), ((1.0,),), (1.0,), False),
        ImplicitTableau("implicit_midpoint", 2, (1 / 2,), ((1 / 2,),), (1.0,), False),
        ImplicitTableau("trapezoid", 2, (0.0, 1.0), ((0.0, 0.0), (1 / 2, 1 / 2)), (1 / 2, 1 / 2), False),
        ImplicitTableau("radauIIA3", 3, (1 / 3, 1.0), ((5 / 12, -1 / 12), (3 / 4, 1 / 4)), (3 / 4, 1 / 4), False),
        # NOTE the reference's gl4 abscissae: both entries are 1/2 - sqrt(3)/6 (fixed_grid_implicit.py:38) — kept
        ImplicitTableau("gl4", 4, (1 / 2 - r3 / 6, 1 / 2 - r3 / 6),
                        ((1 / 4, 1 / 4 - r3 / 6), (1 / 4 + r3 / 6, 1 / 4)), (1 / 2, 1 / 2), False),
        ImplicitTableau("radauIIA5", 5, (2 / 5 - r6 / 10, 2 / 5 + r6 / 10, 1.0),
                        ((11 / 45 - 7 * r6 / 360, 37 / 225 - 169 * r6 / 1800, -2 / 225 + r6 / 75),
                         (37 / 225 + 169 * r6 / 1800, 11 / 45 + 7 * r6 / 360, -2 / 225 - r6 / 75),
                         (4 / 9 - r6 / 36, 4 / 9 + r6 / 36, 1 / 9)),
                        (4 / 9 - r6 / 36, 4 / 9 + r6 / 36, 1 / 9), False),
        ImplicitTableau("gl6", 6, (1 / 2 - r15 / 10, 1 / 2, 1 / 2 + r15 / 10),
                        ((5 / 36, 2 / 9 - r15 / 15, 5 / 36 - r15 / 30),
                         (5 / 36 + r15 / 24, 2 / 9, 5 / 36 - r15 / 24),
                         (5 / 36 + r15 / 30, 2 / 9 + r15 / 15, 5 / 36)),
                        (5 / 18, 4 / 9, 5 / 18), False),
    ]
    g = (2.0 - r2) / 2.0
    tabs.append(ImplicitTableau("sdirk2", 2, (g, 1.0), ((g,), (1 - g, g)), (1 - g, g), True))
    g = 1.0 - r2 / 2.0
    b = r2 / 4.0
    tabs.append(ImplicitTableau("trbdf2", 2, (0.0, 2 * g, 1.0), ((0.0,), (g, g), (b, b, g)), (b, b, g), True))
    return {t.name: t for t in tabs}



IMPLICIT_TABLEAUS = _implicit_tableaus()
