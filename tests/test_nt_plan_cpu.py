"""CPU: host-side plans of the persistent NT GEMMs (no kernel is launched; without a GPU the library plans for 256 CUs).

The column-sum partial rows a caller allocates (vitk_gemm_nt_colsum_rows) must not depend on the CU reserve: FlatGradSink switches the
reserve on and off around every collective, possibly from another thread, between a caller's row query and its launch (the
four-wave kernel used to step aside under a reserve, which changed the row layout; since round 6 it keeps its rows and is launched on
256 - reserve workgroups instead).  The split count of the weight-gradient GEMM, by contrast, is ALLOWED to follow the reserve: it is
passed back into the launch by the caller, so query and launch cannot disagree."""
import pytest

from vit_pytorch_amd import _lib


@pytest.fixture()
def lib():
    lb = _lib.load()
    yield lb
    assert lb.vitk_set_cu_reserve(0) == 0


SHAPES = [(50432, 3072, 768), (50432, 768, 3072), (50432, 2304, 768), (50432, 768, 768), (25216, 4096, 1024), (147712, 5120, 1280), (4096, 512, 256)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_colsum_rows_do_not_depend_on_the_cu_reserve(lib, M, N, K):
    assert lib.vitk_set_cu_reserve(0) == 0
    r0 = lib.vitk_gemm_nt_colsum_rows(M, N, K, N)
    assert r0 > 0
    for c in (8, 32, 64, 192):
        assert lib.vitk_set_cu_reserve(c) == 0 and lib.vitk_get_cu_reserve() == c
        assert lib.vitk_gemm_nt_colsum_rows(M, N, K, N) == r0, (M, N, K, c)


def test_colsum_rows_follow_the_kernel_selection_switch(lib, monkeypatch):
    """VITK_NT_W128=0 (the 8-wave kernel alone: the test suite's A/B switch) changes the rows -- which is why callers query per call."""
    M, N, K = 50432, 3072, 768
    both = lib.vitk_gemm_nt_colsum_rows(M, N, K, N)
    monkeypatch.setenv("VITK_NT_W128", "0")
    alone = lib.vitk_gemm_nt_colsum_rows(M, N, K, N)
    assert both > 0 and alone > 0
    assert both == 2 * (M // 256) or both != alone      # FF1 shape: the four-wave kernel takes every full m-tile (197: 394 rows)


def test_reserve_argument_is_checked_and_the_tn_split_follows_it(lib):
    assert lib.vitk_set_cu_reserve(-1) != 0 and lib.vitk_set_cu_reserve(193) != 0
    assert lib.vitk_set_cu_reserve(0) == 0
    s0 = lib.vitk_gemm_tn_splits(50432, 3072, 768)
    assert lib.vitk_set_cu_reserve(64) == 0
    s1 = lib.vitk_gemm_tn_splits(50432, 3072, 768)
    assert 1 <= s1 <= s0
