"""The installation contract of the reference (/root/reference/makefile:39-47, README.md:55-61: libFL.a, libFL.so -> prefix/lib;
*.mod, *.hpp -> prefix/include; the Python package beside them) reproduced by `make -C fortran-library_amd install prefix=...`,
and the installed tree is SELF-CONTAINED: a caller's HIP objective (include/fl_user_objective.hpp) compiles against
prefix/include alone, a C program links libFL.a, the Python package finds prefix/lib/libFL.so.  CPU only (cross-compiles)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fortran-library_amd")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def prefix(tmp_path_factory):
    if not os.path.exists(os.path.join(PKG, "lib", "libFL.so")):
        pytest.skip("libFL.so not built")
    p = tmp_path_factory.mktemp("prefix")
    subprocess.check_call(["make", "-C", PKG, "install", f"prefix={p}", "-s"], stdout=subprocess.DEVNULL)
    return p


def test_install_lays_out_the_references_artefacts(prefix):
    for rel in ("lib/libFL.so", "lib/libFL.a", "include/fl_nlopt.h", "include/fl_legacy.h", "include/fl_user_objective.hpp",
                "include/FortranLibrary.hpp", "include/NonlinearOptimization.hpp", "include/nonlinearoptimization.mod",
                "include/fortranlibrary.mod", "include/NonlinearOptimization.f90", "include/FortranLibrary.f90",
                "include/fl/fl_solver_launch.hpp", "include/fl/fl_device.hpp", "include/fl/fl_big.hpp",
                "include/fl_user_stream_objective.hpp", "FortranLibrary/__init__.py",
                "FortranLibrary/NonlinearOptimization.py"):
        assert (prefix / rel).exists(), rel
    # nothing in the installed headers reaches back into the source tree
    for f in (prefix / "include").rglob("*.h*"):
        text = f.read_text()
        for fallback in ("fl_solver_launch.hpp", "fl_big.hpp"):  # (the source-tree branch of the __has_include pair)
            text = text.replace('#include "../fortran-library_amd/csrc/%s"' % fallback, "")
        assert "fortran-library_amd/csrc" not in text, f


def test_a_callers_hip_objective_compiles_against_the_installed_headers_alone(prefix, tmp_path):
    src = tmp_path / "caller.hip"
    text = open(os.path.join(ROOT, "tests", "user_objective_caller.hip")).read()
    src.write_text(text.replace('#include "../include/fl_user_objective.hpp"', '#include "fl_user_objective.hpp"'))
    out = tmp_path / "caller"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{prefix}/include", str(src),
                           f"-L{prefix}/lib", "-lFL", f"-Wl,-rpath,{prefix}/lib", "-o", str(out)], cwd=tmp_path)
    assert out.exists()
    # ... and the streaming form for n > 4096 (fl_user_stream_objective.hpp, included twice around the class)
    src2 = tmp_path / "caller_stream.hip"
    text = open(os.path.join(ROOT, "tests", "user_stream_objective_caller.hip")).read()
    assert text.count('#include "../include/fl_user_stream_objective.hpp"') == 2
    src2.write_text(text.replace('#include "../include/fl_user_stream_objective.hpp"', '#include "fl_user_stream_objective.hpp"'))
    out2 = tmp_path / "caller_stream"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{prefix}/include", str(src2),
                           f"-L{prefix}/lib", "-lFL", f"-Wl,-rpath,{prefix}/lib", "-o", str(out2)], cwd=tmp_path)
    assert out2.exists()


def test_a_c_program_links_the_static_library(prefix, tmp_path):
    src = tmp_path / "main.c"
    src.write_text('#include <stdio.h>\n#include "fl_nlopt.h"\nint main(void){fl_options o; fl_default_options(&o, FL_SOLVER_LBFGS);'
                   'int t, e; fl_reduction_geometry(1024, &t, &e); printf("%d %d %d %d\\n", fl_version(), o.memory, t, e); return 0;}\n')
    obj = tmp_path / "main.o"
    subprocess.check_call(["gcc", "-c", f"-I{prefix}/include", str(src), "-o", str(obj)])
    exe = tmp_path / "main_static"
    subprocess.check_call([HIPCC, str(obj), str(prefix / "lib" / "libFL.a"), "-ldl", "-lpthread", "-o", str(exe)])
    assert subprocess.check_output([str(exe)]).decode().split() == ["104", "10", "128", "8"]


def test_the_installed_python_package_finds_the_installed_library(prefix):
    code = ("import sys; sys.path.insert(0, %r); import FortranLibrary.basic as B; import FortranLibrary.NonlinearOptimization as N; "
            "assert B.library_path().startswith(%r), B.library_path(); print(N.FL.fl_version())" % (str(prefix), str(prefix)))
    env = {k: v for k, v in os.environ.items() if k not in ("FL_LIBRARY", "PYTHONPATH")}
    assert subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip() == "104"
