"""The product's Butcher tableaus are bit-identical to the reference's fp64 tensors."""
import numpy as np
import pytest

from _cases import load
from torchdiffeq_amd.tableaus import ADAPTIVE_HEUN, BOSH3, DOPRI5, DOPRI8, FEHLBERG2, TSIT5, SparseRow


ALL = [DOPRI5, DOPRI8, TSIT5, BOSH3, FEHLBERG2, ADAPTIVE_HEUN]


@pytest.mark.parametrize("tab", ALL, ids=[t.name for t in ALL])
def test_tableau_matches_reference_bits(tab):
    z = load("tableaus.npz")
    alpha, beta, c_sol, c_err, c_mid = tab.dense()
    assert np.array_equal(alpha, z[f"{tab.name}_alpha"])
    assert np.array_equal(np.concatenate(beta), z[f"{tab.name}_beta_flat"])
    assert np.array_equal(c_sol, z[f"{tab.name}_c_sol"])
    assert np.array_equal(c_err, z[f"{tab.name}_c_error"])
    assert np.array_equal(c_mid, z[f"{tab.name}_c_mid"])
    # rk_common.py:83 shortcut: holds for the Dormand–Prince pairs and bosh3; tsit5 / fehlberg2 / adaptive_heun
    # take the extra solution combine
    assert tab.fsal_solution == (tab.name in ("dopri5", "dopri8", "bosh3"))


def test_structural_zero_counts():
    """Non-zeros per row = the algorithmic words of SURVEY.md §8(d)."""
    assert [len(r.idx) for r in DOPRI5.beta_rows()] == [1, 2, 3, 4, 5, 5]
    assert [len(r.idx) for r in DOPRI8.beta_rows()] == [1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 9, 9, 9]
    assert len(SparseRow.from_dense(DOPRI5.c_error).idx) == 6
    assert len(SparseRow.from_dense(DOPRI8.c_error).idx) == 9
    assert len(SparseRow.from_dense(DOPRI5.c_mid).idx) == 6
    assert len(SparseRow.from_dense(DOPRI8.c_mid).idx) == 10
