import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """Built artefacts are git-ignored: a fresh checkout compiles them here, once
    (hipcc cross-compiles gfx950 without a GPU; ~15 s)."""
    need = [os.path.join(ROOT, "verifybamid_amd", "libvb2.so"),
            os.path.join(ROOT, "verifybamid_amd", "bin", "VerifyBamID"),
            os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "oracle", "_check_exp.bin"),
            os.path.join(ROOT, "tests", "stub_rccl", "librccl_stub.so")]
    # ... and rebuilds them whenever a source is newer than the oldest artefact: a stale binary
    # must not mask a broken source file (make itself only recompiles what changed)
    srcs = []
    for d in ("verifybamid_amd/csrc", "include", "oracle", "tests/stub_rccl"):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".cpp", ".hip", ".inc", ".h", ".c", "Makefile")):
                srcs.append(os.path.join(ROOT, d, f))
    if all(os.path.exists(p) for p in need):
        oldest = min(os.path.getmtime(p) for p in need)
        if all(os.path.getmtime(f) <= oldest for f in srcs):
            return
    import __graft_entry__
    __graft_entry__.build()


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def tunable():
    """tunable(name, value) sets a run-time switch of the library (csrc/tunables.h) for the rest of the test."""
    from verifybamid_amd import _abi
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = _abi.get_tunable(name)
        _abi.set_tunable(name, value)
    yield set_
    for name, value in saved.items():
        _abi.set_tunable(name, value)


def fixture_input_must_match(got_sha, want_sha, what):
    """A committed fixture is only evidence if the seeded generator reproduces the input it was made from.
    A mismatch (another numpy drawing other samples) FAILS -- a silent skip would let the at-size
    parity evidence evaporate -- unless VB2_ALLOW_FIXTURE_DRIFT=1 says the box is known to differ."""
    import pytest
    if got_sha == want_sha:
        return
    msg = ("%s: the seeded generator produced input %s..., the committed fixture was made from %s... "
           "(regenerate with tests/golden/make_fixtures.py, or set VB2_ALLOW_FIXTURE_DRIFT=1 to skip)"
           % (what, got_sha[:12], want_sha[:12]))
    if os.environ.get("VB2_ALLOW_FIXTURE_DRIFT", "") == "1":
        pytest.skip(msg)
    pytest.fail(msg)
