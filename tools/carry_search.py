"""Search for the carried-partial-sum assignment (target row -> host row) of a tableau that minimises the words moved
per element and trial step (torchdiffeq_amd/tableaus.py `_CARRY_HOSTS`; cost model = `tableaus._plan_from_hosts`).

    python tools/carry_search.py [dopri5 dopri8 tsit5 bosh3 fehlberg2 adaptive_heun]

Simulated annealing over the assignments with hosts that are formed in full, <= 4 outputs per launch and at most two
error terms left to the norm kernel; prints the best assignment found next to the row-by-row word count.  Needs no GPU."""
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchdiffeq_amd import tableaus as tb  # noqa: E402


def cost(tab, hosts):
    try:
        plan = tb._plan_from_hosts(tab, dict(hosts))
    except (AssertionError, KeyError):
        return None
    return plan.words, plan.launches


def search(tab, seeds=8, iters=40000):
    S = len(tab.beta)
    R = S if tab.fsal_solution else S + 1
    base = {R: R - 1}                      # the end-of-step fusion alone: the last launch row hosts the error
    best = (cost(tab, base), dict(base))
    for seed in range(seeds):
        rnd = random.Random(seed)
        cur, cc, T = dict(base), cost(tab, base), 1.0
        for _ in range(iters):
            t = rnd.randrange(1, R + 1)
            cand = dict(cur)
            h = rnd.choice([None] + list(range(1, t)))
            if h is None:
                if t == R:
                    continue               # the error always has a host
                cand.pop(t, None)
            else:
                if cand.get(h) is not None or any(v == t for v in cand.values()):
                    continue
                cand[t] = h
            c = cost(tab, cand)
            if c is None:
                continue
            if c <= cc or rnd.random() < math.exp(-(c[0] - cc[0]) / T):
                cur, cc = cand, c
                if cc < best[0]:
                    best = (cc, dict(cur))
            T = max(0.05, T * 0.9998)
    return best


for name in (sys.argv[1:] or ["dopri5", "dopri8", "tsit5", "bosh3", "fehlberg2", "adaptive_heun"]):
    tab = tb.ADAPTIVE_TABLEAUS[name]
    (words, launches), hosts = search(tab)
    print(f"{name}: row by row {tb.row_by_row_words(tab)} words; best found {words} words in {launches} launches with "
          f"hosts {dict(sorted(hosts.items()))}; shipped: {tb._CARRY_HOSTS.get(name)}")
