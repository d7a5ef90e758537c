"""C++ tests of the host mirror (tests/cpp): the reference's ModalSolverTest / ModalRenderTest / ContactModelTest
properties, written against the mirrored headers under the reference's include paths.  On the CPU: they compile and
link (the API surface is intact) and the scalar contact model runs; on the GPU box: all three run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "mesheditor_amd", "libmodalhost.so")):
        ge.build()
    subprocess.run(["make", "-s", "-C", CPP], check=True)


def _run(name, timeout=600):
    p = subprocess.run([os.path.join(CPP, "bin", name)], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_cpp_tests_compile_against_the_mirror():
    _build()
    for name in ("modal_solver_test", "modal_render_test", "contact_model_test"):
        assert os.path.exists(os.path.join(CPP, "bin", name))


def test_contact_model_known_answers():
    _build()
    assert "0 failure(s)" in _run("contact_model_test")


@pytest.mark.gpu
def test_modal_render_properties_cpp():
    _build()
    assert "0 failure(s)" in _run("modal_render_test")


@pytest.mark.gpu
def test_modal_solver_closed_forms_cpp():
    _build()
    assert "0 failure(s)" in _run("modal_solver_test")
