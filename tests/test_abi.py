"""CPU-side checks of the drop-in boundary: the C ABI library builds, loads and exports exactly the
symbols include/pnpp_hip.h declares; the Python surface keeps the reference's names and state_dict
keys; and the product path refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, PKG


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pnpp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pnpp_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def libpath():
    from pnpp_hip import build
    return build.build()


def test_header_symbols_exported(libpath):
    h = ctypes.CDLL(libpath)
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/pnpp_hip.h but not exported"


def test_binding_table_matches_header(libpath):
    from pnpp_hip import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert _lib.lib().pnpp_abi_version() == 5


def test_shipped_library_has_no_experiment_switches(libpath):
    """The kernel sources carry compile-time timing experiments that compute WRONG results (WSP_EXP, WSQ_EXP, WSX_EXP, FCF_EXP,
    PNPP_WS_EXP_NO_MFMA, WSQ_PLAIN) and in-kernel stamps (PNPP_STAMPS); the library that ships is built with none of them."""
    from pnpp_hip import _lib
    assert _lib.lib().pnpp_build_flags() == 0


def test_struct_layouts_match_c():
    """ctypes mirrors of the descriptor structs have the size the C compiler gives them."""
    import subprocess, tempfile, textwrap
    from pnpp_hip import _lib
    src = textwrap.dedent('''
        #include <stdio.h>
        #include "pnpp_hip.h"
        int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(pnpp_sa_desc), sizeof(pnpp_sa_fwd_args),
            sizeof(pnpp_sa_bwd_args), sizeof(pnpp_fc_desc), sizeof(pnpp_fc_fwd_args), sizeof(pnpp_fc_bwd_args)); return 0; }
    ''')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mine = [ctypes.sizeof(t) for t in (_lib.SaDesc, _lib.SaFwdArgs, _lib.SaBwdArgs, _lib.FcDesc, _lib.FcFwdArgs, _lib.FcBwdArgs)]
    assert sizes == mine


def test_argument_errors_without_gpu(libpath):
    """Argument validation happens before any launch, so it can be exercised on a CPU-only box."""
    from pnpp_hip import _lib
    h = _lib.lib()
    assert h.pnpp_knn(None, None, 1, 1, 1, 1, None, None) == _lib.PNPP_ERR_ARG
    assert b"null" in h.pnpp_last_error()
    assert h.pnpp_knn(8, 8, 1, 4, 5, 6, 8, None) == _lib.PNPP_ERR_RANGE          # k > N, like torch.topk
    assert h.pnpp_knn(8, 8, 1, 4, 500, 200, 8, None) == _lib.PNPP_ERR_ARG        # nsample over the kernel maximum
    d = _lib.SaDesc()
    d.B, d.N, d.S, d.K, d.D, d.L = 2, 64, 8, 4, 0, 3
    d.C[0], d.C[1], d.C[2] = 32, 48, 64                                           # 48 is not a multiple of 32
    assert h.pnpp_sa_saved_bytes(ctypes.byref(d)) == 0
    assert b"multiple of 32" in h.pnpp_last_error()
    d.C[1] = 64
    assert h.pnpp_sa_saved_bytes(ctypes.byref(d)) > 0 and h.pnpp_sa_scratch_bytes(ctypes.byref(d)) > 0
    with pytest.raises(ValueError):
        _lib.check(_lib.PNPP_ERR_ARG)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.PNPP_ERR_RANGE)


def test_python_surface_names_and_state_dict():
    import models
    from models import base
    from models.pointnet_pp_8dir import PointNetSetAbstraction, PointNetPP8Dir, DIRS_8
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from models.pointnet_pp_mvM import PointNetPPMvM, mvm_density_on_grid, _as_points_last
    assert callable(base.index_points) and callable(base.square_distance) and callable(base.query_ball_point)
    assert DIRS_8.shape == (8, 3)
    counts = {PointNetPPVonMises: 1465922, PointNetPPMvM: 1469520, PointNetPP8Dir: 1467464}
    for cls, n in counts.items():
        m = cls()
        assert sum(p.numel() for p in m.parameters()) == n
        sd = m.state_dict()
        for s, cin in (("sa1", 3), ("sa2", 131), ("sa3", 259)):
            assert sd[f"{s}.convs.0.weight"].shape[1:] == (cin, 1, 1)
            for l in range(3):
                for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
                    assert f"{s}.bns.{l}.{k}" in sd
    sd = PointNetPPMvM().state_dict()
    assert sd["head_pi.weight"].shape == (4, 256) and sd["head_mu.weight"].shape == (8, 256)
    assert sd["head_pi.weight"].abs().sum() == 0 and sd["head_mu.weight"].abs().sum() == 0
    assert "ln1.weight" in sd and "ln2.bias" in sd
    with pytest.raises(ValueError):
        _as_points_last(torch.zeros(2, 5, 7))
    with pytest.raises(AssertionError):
        _as_points_last(torch.zeros(5, 3))
    assert _as_points_last(torch.zeros(2, 5, 3)).shape == (2, 5, 3) and _as_points_last(torch.zeros(2, 3, 5)).shape == (2, 5, 3)
    th, p = mvm_density_on_grid(torch.zeros(2, 4), torch.ones(2, 4), torch.full((2, 4), 0.25), num=90)
    assert th.shape == (89,) and p.shape == (2, 89) and torch.allclose(p.sum(-1), torch.ones(2), atol=1e-5)


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU rather than compute something else."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from models import base
    m = PointNetPPVonMises()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(2, 64, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        base.square_distance(torch.rand(1, 4, 3), torch.rand(1, 5, 3))


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/:
    nothing under the package imports, includes, loads or executes it."""
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|#include\s*[\"<][^\">]*oracle|liboracle|oracle[/.]restatement"
                     r"|oracle/_build", re.M)
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not bad.search(text), (dirpath, f)
