import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the suite needs the built artefacts (git-ignored): the engine library + CLI and the test-only checkers under oracle/.
    # Both steps are no-ops when everything is up to date (as on the GPU box, where the snapshot carries the built files).
    import subprocess
    try:
        from pansvr_amd import build as b
        b.build(force=False, verbose=False)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    except Exception as e:  # a missing compiler shows up as the individual tests' own failures
        sys.stderr.write("conftest: build step failed: %r\n" % (e,))
