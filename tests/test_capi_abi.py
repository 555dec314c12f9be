"""CPU: the C-ABI library loads and exports every symbol include/mvs_hip.h declares; without a GPU the
product path fails loudly instead of falling back."""
import os
import re

import pytest

from metagenome_vector_sketches_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    with open(os.path.join(ROOT, "include", "mvs_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvs_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _capi.load_library()
    declared = _header_symbols()
    assert len(declared) >= 20
    bound = {name for name, _, _ in _capi.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), name
        assert name in bound, "no ctypes signature for " + name
    assert b"gfx950" in lib.mvs_version()


def test_pure_host_helpers_match_reference_formulas():
    # src/pairwise_comp_optimized.cpp:903-906, :938-940
    assert _capi.chunk_size(12, 2048) == 192
    assert _capi.chunk_size(12, 4096) == 48
    assert _capi.shard_rows(61, 2, 0) == (0, 31)
    assert _capi.shard_rows(61, 2, 1) == (31, 61)
    assert _capi.shard_rows(100, 8, 7) == (91, 100)
    assert _capi.shard_rows(3, 8, 7) == (3, 3)
    assert [_capi.limbs_for_max_abs(x) for x in (0, 127, 128, 8127, 8128, 32639, 32640, 8355711, 8355712, 2**31)] == \
        [1, 1, 2, 2, 2, 2, 3, 3, 4, 4]


def test_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_capi.MvsError) as ei:
        _capi.Context(0)
    assert ei.value.code == _capi.MVS_E_HIP


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "metagenome_vector_sketches_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert "pyoracle" not in src and "mvs_oracle" not in src and "libmvs_oracle" not in src, fn


def test_library_reads_the_environment_only_when_a_context_is_created():
    """tuning switches live in the context (mvs_ctx_set_option); below the C ABI exactly one getenv exists, in the
    function that seeds a new context's options, and the k-loop ablations are not part of the product library"""
    csrc = os.path.join(ROOT, "metagenome_vector_sketches_amd", "csrc")
    hits = []
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(csrc, fn)) as f:
                for i, line in enumerate(f, 1):
                    if "getenv(" in line:
                        hits.append((fn, i, line.strip()))
    assert len(hits) == 1 and hits[0][0] == "mvs_capi.hip", hits
    with open(os.path.join(csrc, "mvs_capi.hip")) as f:
        text = f.read()
    assert text.index("void options_from_env") < text.index("getenv(") < text.index("int check_kernel")
    lib = _capi.load_library()
    assert not hasattr(lib, "mvs_ctx_debug")                       # no debug entry points
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "mvs_ctx_set_option" in syms and "mvs_comm_create" in syms and "mvs_allgather_planes" in syms


def test_generated_code_of_the_hand_counted_loads():
    """`make check-isa`: the direct-B ping-pong kernels (k_pairwise_pp<..., BD = 1>) issue their B-fragment loads as inline asm
    and count vmcnt by hand; the compiler believes the results are there at once.  tools/check_isa.py disassembles the gfx950
    code object inside the library that ships and checks that nothing touches a loaded fragment register before the wait that
    retires it and that every MFMA's B operand is such a register -- and that it would notice: three mutated variants of the
    real code (a register copy right after the load, a foreign B operand, weakened waits) must be rejected."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), "--self-test"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "direct-B kernels pass" in r.stdout and "all 3 mutations" in r.stdout
