import atexit, faulthandler, json, os, subprocess, sys, threading
import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
sys.path.insert(0, os.path.join(_HERE, ".."))

# ---- children of the suite come from a helper process that was started HERE, before any test could have made a GPU call (tests/childserver.py) -------------
_srv = subprocess.Popen([sys.executable, os.path.join(_HERE, "childserver.py")], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
_srv_lock = threading.Lock()


def _stop_server():
    try:
        _srv.stdin.close(); _srv.wait(timeout=5)
    except Exception:   # noqa: BLE001
        pass
atexit.register(_stop_server)


class ChildResult:
    def __init__(self, d):
        self.returncode = d["returncode"]; self.stdout = d["stdout"]; self.stderr = d["stderr"]; self.timed_out = d.get("timeout", False)


def run_child(argv, env=None, cwd=None, timeout=None, input=None):
    """subprocess.run(argv, capture_output=True, text=True, ...) by way of the fork-server: the pytest process itself never forks once it holds a GPU"""
    req = {"argv": [str(a) for a in argv], "env": dict(env) if env is not None else dict(os.environ), "cwd": cwd, "timeout": timeout, "input": input}
    with _srv_lock:
        _srv.stdin.write(json.dumps(req) + "\n"); _srv.stdin.flush()
        line = _srv.stdout.readline()
    if not line:
        raise RuntimeError("tests/childserver.py is gone")
    r = ChildResult(json.loads(line))
    if r.timed_out:
        raise subprocess.TimeoutExpired(argv, timeout, output=r.stdout, stderr=r.stderr)
    return r


# ---- the native last words of a GPU test process (an abort whose message pytest's fd capture swallowed) ------------------------------------
# On a box with a GPU: Python tracebacks of every thread on SIGABRT / SIGSEGV / SIGBUS go to gpurun_out/native_stderr.log, and while a test's call phase runs,
# fd 2 itself points at that file (inside pytest's own capture, so what the runtime, the C library or the product print just before abort() survives the process).
_GPU_BOX = os.path.exists("/dev/kfd") and not os.environ.get("EZHIP_TESTS_NO_STDERR_LOG")
_log_fd = None
if _GPU_BOX:
    try:
        _logdir = os.path.join(_HERE, "..", "gpurun_out")
        os.makedirs(_logdir, exist_ok=True)
        _log = open(os.path.join(_logdir, "native_stderr.log"), "a", buffering=1)
        _log.write("==== pytest process %d ====\n" % os.getpid())
        faulthandler.enable(file=_log, all_threads=True)
        _log_fd = _log.fileno()
    except OSError:
        _log_fd = None


@pytest.hookimpl(hookwrapper=True, trylast=True)
def pytest_runtest_call(item):
    if _log_fd is None or "capfd" in getattr(item, "fixturenames", ()):
        yield
        return
    os.write(_log_fd, ("-- %s\n" % item.nodeid).encode())
    saved = os.dup(2)
    try:
        sys.stderr.flush()
    except Exception:   # noqa: BLE001
        pass
    os.dup2(_log_fd, 2)
    try:
        yield
    finally:
        os.dup2(saved, 2); os.close(saved)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: test needs oracle/_ref/libezref.so (reference build)")


# Host-only assertions (grid table, set-up math, options: SURVEY 8 rows a2 - a7, a24) need no device, but the round-end driver only runs `-m gpu` on the GPU box:
# a test decorated with @both_legs is collected twice -- once unmarked (the CPU suite here) and once marked gpu (the driver's suite there) -- and takes `leg`.
both_legs = pytest.mark.parametrize("leg", ["cpu_suite", pytest.param("gpu_suite", marks=pytest.mark.gpu)])
