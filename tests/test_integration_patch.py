"""integration/qbhip_reference.patch applies cleanly to the unchanged reference (build container only: the GPU box has
no /root/reference, and nothing of the reference is committed)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
PATCH = os.path.join(ROOT, "integration", "qbhip_reference.patch")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")
def test_patch_applies_to_the_unchanged_reference():
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "src"))
        for f in ("sparse.cc", "lanczos.cc", "qbasis.h"):
            shutil.copy(os.path.join(REF, f), os.path.join(tmp, "src", f))
        before_h = open(os.path.join(tmp, "src", "qbasis.h")).read()
        dry = subprocess.run(["patch", "--dry-run", "-p1", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
        assert dry.returncode == 0, dry.stdout + dry.stderr
        assert "FAILED" not in dry.stdout and "fuzz" not in dry.stdout
        real = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
        assert real.returncode == 0, real.stdout + real.stderr
        sp = open(os.path.join(tmp, "src", "sparse.cc")).read()
        lz = open(os.path.join(tmp, "src", "lanczos.cc")).read()
        assert open(os.path.join(tmp, "src", "qbasis.h")).read() == before_h            # the public header is untouched
        # every binding of INTEGRATION.md is in place, and no MKL sparse call is left on the complex path
        for needle in ("qbh_csr_create(", "qbh_csr_destroy(", "qbh_multmv2("):
            assert needle in sp
        assert sp.count("create_handle(&handle, dim, nnz, sym, ia, ja, val)") == 2
        assert "mkl_sparse_z_mv" not in sp and "mkl_sparse_z_create_csr" not in sp
        for needle in ("qbh_lanczos(", "qbh_lanczos_ckpt(", "qbh_eigenvec_cg("):
            assert needle in lz
        # the explicit instantiations for model<T> as MAT are still there (not accelerated, must still link)
        assert "const model<std::complex<double>> &mat, std::complex<double> v[]," in lz


def test_patch_only_touches_the_two_translation_units():
    files = [ln.split()[1] for ln in open(PATCH) if ln.startswith("+++ ")]
    assert sorted(os.path.basename(f) for f in files) == ["lanczos.cc", "sparse.cc"]
    assert all("qbasis.h" not in ln for ln in open(PATCH) if ln.startswith(("+++", "---")))
