"""CPU tests of bench.py's record keeping: the fields of the JSON line that are read from committed files must come from the
profile of the SAME kernel and geometry (round 5's line took the 3 x 84 x 84 kernel's traffic for the dSprites headline)."""
import json
import os
import re
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def _traffic(path):
    m = re.search(r'== HBM traffic \(JSON\) ==\n(\{.*\})', open(path).read())
    t = json.loads(m.group(1))['k_dec_b']
    return t['hbm_read_bytes_per_image'] + t['hbm_write_bytes_per_image']


def test_committed_traffic_is_selected_by_geometry(tmp_path, monkeypatch):
    import bench
    bpi, src = bench.committed_traffic('k_dec_b', 'dsprites')
    assert re.fullmatch(r'r\d+_v\d+_rocprof_summary\.txt', src), src                 # never an _ai_ / _b3_ profile
    assert bpi == _traffic(os.path.join(ROOT, 'profiles', src))
    # k_dec_b4 reads y2 (256 KiB per image) + halo rows and writes a sum (+ every third image): within 1.0 - 1.2 x algorithmic
    assert 1.0 <= bpi / bench.ALG_BYTES_DECB_IMAGE <= 1.2, bpi
    bpa, srca = bench.committed_traffic('k_dec_b', 'animalai')
    assert re.fullmatch(r'r\d+_v\d+_ai_rocprof_summary\.txt', srca), srca
    assert bpa == _traffic(os.path.join(ROOT, 'profiles', srca)) and bpa > 1.5 * bpi   # 84 x 84: y2 is 451 KB per image
    # the choice is by NAME PATTERN, not by age: a newer _ai_ / _b3_ file next to an older headline profile changes nothing
    prof = tmp_path / 'profiles'
    prof.mkdir()
    body = '== HBM traffic (JSON) ==\n{"k_dec_b": {"hbm_read_bytes_per_image": %d, "hbm_write_bytes_per_image": 1}}\n'
    (prof / 'r6_v1_rocprof_summary.txt').write_text(body % 100)
    (prof / 'r6_v2_rocprof_summary.txt').write_text(body % 200)
    (prof / 'r6_v10_rocprof_summary.txt').write_text('no traffic section in this one\n')
    (prof / 'r6_v3_ai_rocprof_summary.txt').write_text(body % 300)
    (prof / 'r6_v4_b3_rocprof_summary.txt').write_text(body % 400)
    (prof / 'r7_v1_ai_rocprof_summary.txt').write_text(body % 500)
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench.committed_traffic('k_dec_b', 'dsprites') == (201, 'r6_v2_rocprof_summary.txt')
    assert bench.committed_traffic('k_dec_b', 'animalai') == (501, 'r7_v1_ai_rocprof_summary.txt')
    assert bench.committed_traffic('k_other', 'dsprites') == (None, None)


def test_rank_count_is_labelled_by_backend():
    import bench

    class FakeDist:
        def __init__(self, be):
            self.be = be

        def get_world_size(self):
            return 8

        def get_backend(self):
            return self.be
    import torch
    for be, key in (('nccl', 'rccl_ranks'), ('gloo', 'ranks')):
        rk = bench.Ranks(FakeDist(be), 8, 0, torch.device('cpu'), be)
        info = rk.info()
        assert info == {key: 8, 'backend': be}


def test_split_mode_traffic_comes_from_that_modes_profile():
    """the experiment legs' roofline.traffic: bytes per launch of the persistent split kernel from the newest committed profile of the SAME mode"""
    import bench
    for opt, tag in (('mfma_bf16x3', 'b3'), ('mfma_f16x2', 'f16')):
        b, src = bench.committed_split_traffic('k_dec_b_b3', opt)
        assert re.fullmatch(rf'r\d+_v\d+_{tag}_rocprof_summary\.txt', src), src
        # ConvT3 reads y2 (19 200 x 256 KiB = 5.03 GB) once -- halo rows mostly from L2 -- and writes sums + every third image
        assert 5.0e9 <= b <= 7.0e9, b
        a_, _ = bench.committed_split_traffic('k_dec_a_b3', opt)
        assert 6.0e9 <= a_ <= 7.5e9, a_              # x4 in (1.26 GB) + weights + y2 out (5.03 GB)
    assert bench.committed_split_traffic('k_nothing', 'mfma_f16x2') == (None, None)
