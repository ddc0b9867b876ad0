"""The C-ABI library loads and exports every symbol include/lzgpu.h declares; host-only entry
points work without a GPU; device entry points fail LOUDLY (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import torch

from oracle import lzo
from lastz_amd import lzgpu
import helpers as H


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(lzgpu.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return lzgpu.Lib()


def test_exports_match_header(lib):
    hdr = open(os.path.join(H.ROOT, "include", "lzgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(lzgpu_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(lzgpu.EXPORTS)
    for name in declared:
        assert hasattr(lib.L, name), name


def test_struct_layouts_match_header(lib):
    # sizes the C compiler gives the PODs (guards the ctypes mirrors)
    src = '#include "lzgpu.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(lz_seed_desc),sizeof(lz_table_geom),sizeof(lz_search_args),sizeof(lz_hsp),' \
          'sizeof(lz_gapped_args),sizeof(lz_align),sizeof(lz_counters));return 0;}'
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(H.ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [C.sizeof(lzgpu.SeedDesc), C.sizeof(lzgpu.TableGeom), C.sizeof(lzgpu.SearchArgs),
                     lzgpu.HSP_DTYPE.itemsize, C.sizeof(lzgpu.GappedArgs), lzgpu.ALIGN_DTYPE.itemsize,
                     C.sizeof(lzgpu.Counters)]


@pytest.mark.parametrize("pattern,wt", [(H.DEFAULT_SEED, 1), ("11111111", 0), ("1110101100110010101111", 1),
                                        ("111101101111", 2), ("0011x1100", 1)])
def test_seed_compiler_matches_oracle(lib, pattern, wt):
    a, b = lib.seed(pattern, wt), lzo.seed(pattern, wt)
    assert (a.length, a.weight_bits, a.num_parts, a.num_probes) == (b.length, b.weight, b.num_parts, b.num_probes)
    assert [(a.shift[i], a.mask[i]) for i in range(a.num_parts)] == [(b.shift[i], b.mask[i]) for i in range(b.num_parts)]
    assert [a.probe_xor[i] for i in range(a.num_probes)] == [b.probe_xor[i] for i in range(b.num_probes)]


def test_halfweight_seed_is_declined(lib):
    with pytest.raises(lzgpu.NotHandled):
        lib.seed("1T1T1T1T11", 0)


@pytest.mark.skipif(torch.cuda.is_available(), reason="this checks the no-GPU behaviour")
def test_no_silent_cpu_fallback(lib):
    assert lib.probe() == -1                                # LZGPU_ERR_NO_DEVICE
    t = np.frombuffer(b"ACGT" * 100, dtype=np.uint8)
    with pytest.raises(lzgpu.LzGpuError):
        lib.table_prepare(t, lib.seed(), lzo.upper_nuc_to_bits())
    _, masked = H.scoring()
    with pytest.raises(lzgpu.LzGpuError):
        lib.seed_hit_search(masked, q=t)
