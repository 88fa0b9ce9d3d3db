"""CPU: host logic of cloops_amd.saturation (scripts/jd2saturation: the draw of the samples, the script's own dispatch rules, the
per-eps filter + combine, minPts scaled by the depth, the `min([])` failure) against golden vectors made by the script's own
functions, with the GPU replaced by the oracle-backed FakeChromosome."""
import numpy as np
import pytest

import fake_backend
import saturation_checks
from cloops_amd import api, pipe, saturation


@pytest.fixture()
def cpu_backend(monkeypatch):
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    yield
    pipe.CACHE.clear()


def test_flow_against_the_scripts_own_functions(cpu_backend):
    saturation_checks.check_flow(pipe, saturation, cuts=(0, 3000))


def test_overlap_rule():
    """cLoops/bk.py:1-19 (the function scripts/jd2saturation:25 means to import)"""
    a = ["c", 100, 200, "c", 1000, 1100]
    assert saturation.checkOverlap(a, ["c", 200, 300, "c", 1100, 1200])          # touching ends overlap (<=)
    assert saturation.checkOverlap(a, ["c", 50, 500, "c", 1050, 1060])           # containment either way
    assert not saturation.checkOverlap(a, ["c", 201, 300, "c", 1000, 1100])
    assert not saturation.checkOverlap(a, ["c", 100, 200, "c", 1101, 1200])


def test_saturation_table(tmp_path, cpu_backend):
    """getSets / getSaturation on hand-made `.loop` tables: share of the full data's significant loops a sample recovers"""
    import pandas as pd

    def table(path, rows):
        pd.DataFrame({"iva": ["chr1:%d-%d" % (a, b) for a, b, _, _, _ in rows], "ivb": ["chr1:%d-%d" % (c, d) for _, _, c, d, _ in rows],
                      "significant": [float(s) for _, _, _, _, s in rows]}, index=["l%d" % k for k in range(len(rows))]).to_csv(path, sep="\t", index_label="loopId")
    full = str(tmp_path / "full.loop")
    table(full, [(100, 200, 1000, 1100, 1), (300, 400, 5000, 5100, 1), (900, 950, 9000, 9100, 0), (2000, 2100, 8000, 8100, 1)])
    s1 = str(tmp_path / "depth_0.5_rep_0.loop")
    table(s1, [(150, 160, 1050, 1060, 1), (300, 400, 5000, 5100, 0)])
    s2 = str(tmp_path / "depth_0.5_rep_1.loop")
    table(s2, [(150, 160, 1050, 1060, 1), (390, 500, 5100, 5200, 1), (2000, 2100, 8000, 8100, 1)])
    ds = saturation.getSaturation(full, [s1, s2], str(tmp_path / "out"))
    assert np.isclose(ds.loc[0, 0.5], 100.0 / 3) and np.isclose(ds.loc[1, 0.5], 100.0)
    assert (tmp_path / "out_ResamplingRatios.txt").exists()
