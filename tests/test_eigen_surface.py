"""The Eigen-typed class surface (VERDICT r1 missing #6): with <Eigen/Core> on the include path the host mirror declares
computeNeighborhoodDistribution / searchNeighbors with exactly the reference's signatures
(include/lioOptimization.h:340-343).  The image has no Eigen, so the branch is compiled against a minimal stand-in
(tests/stub_eigen); the test pins the member-function-pointer types.  CPU only, compile-time."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "sr_livo_amd/csrc/host/lioOptimization.h"
#ifndef SRL_HAVE_EIGEN
#error "the __has_include(<Eigen/Core>) branch was not taken"
#endif
using namespace srlivo;
typedef std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> EigenList;
// include/lioOptimization.h:340
static NeighborhoodEigen (lioOptimization::*p_cnd)(const EigenList &) = &lioOptimization::computeNeighborhoodDistribution;
// include/lioOptimization.h:342-343
static EigenList (lioOptimization::*p_sn)(voxelHashMap &, const Eigen::Vector3d &, int, double, int, int, std::vector<voxel> *) = &lioOptimization::searchNeighbors;
// the three members whose signatures carry no Eigen type keep the reference's spelling (:334-338)
static optimizeSummary (lioOptimization::*p_opt)(cloudFrame *, const icpOptions &, double) = &lioOptimization::optimize;
static optimizeSummary (lioOptimization::*p_bpr)(const icpOptions &, voxelHashMap &, std::vector<point3D> &, std::vector<planeParam> &, cloudFrame *, double &) = &lioOptimization::buildPlaneResiduals;
static optimizeSummary (lioOptimization::*p_upd)(const icpOptions &, voxelHashMap &, std::vector<point3D> &, cloudFrame *) = &lioOptimization::updateIEKF;
int main() { return (p_cnd && p_sn && p_opt && p_bpr && p_upd) ? 0 : 1; }
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("eigen_dir", [os.path.join("tests", "stub_eigen"), os.path.join("oracle", "ref_shim")])
def test_reference_signatures_compile_with_eigen_on_the_include_path(tmp_path, eigen_dir):
    """eigen_dir: the minimal stand-in of this test, and the fuller stand-in Eigen the reference's own translation units are
    compiled against (oracle/ref_shim/Eigen/Core) -- the same header must serve both."""
    src = tmp_path / "surface.cpp"
    src.write_text(SRC)
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-I", ROOT, "-I", os.path.join(ROOT, eigen_dir), "-I", "/opt/rocm/include", str(src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # and without Eigen on the path the same header still compiles (the srl:: surface only)
    src2 = tmp_path / "plain.cpp"
    src2.write_text('#include "sr_livo_amd/csrc/host/lioOptimization.h"\n#ifdef SRL_HAVE_EIGEN\n#error "unexpected Eigen"\n#endif\nint main() { return 0; }\n')
    r2 = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", ROOT, "-I", "/opt/rocm/include", str(src2)], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr[-3000:]
