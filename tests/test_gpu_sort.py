"""The radix sort on its own (kernels/radix_sort.hip through tools/sortbench.hip): every tile shape against
std::stable_sort on 360 cases -- sizes around the tile borders (1 .. 5 000 011 items), uniform / skewed / constant digits,
u32 and u64 keys with and without payload, bit windows [lo, hi), the split last pass of the k-mer index -- plus the
look-back's time-out word.  Replaces dalign/filter.c:230-435 (lex_sort), whose order is the stable order on the same bits."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "damar_amd", "bin", "sortbench")


@pytest.mark.gpu
@pytest.mark.parametrize("threads,lb64", [(1024, False), (512, False), (256, False), (256, True)])
def test_gpu_radix_sort_tile_shapes_equal_stable_sort(threads, lb64):
    """lb64: the 64-bit look-back words that sorts of 2^30 items and more use (256 x 16 tiles), forced at test sizes."""
    assert os.path.exists(EXE), "damar_amd/bin/sortbench is built by damar_amd/csrc/Makefile"
    env = dict(os.environ, DAMAR_SORT_THREADS=str(threads))
    if lb64:
        env["DAMAR_SORT_LB64"] = "1"
    r = subprocess.run([EXE, "check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "check ok: 360 cases (threads %d" % threads in r.stdout, r.stdout[-500:]
