"""TEST INFRASTRUCTURE: a fork-server for the GPU suite.

tests/conftest.py starts this file as a helper process when it is imported -- before any test has made a GPU call -- and every child interpreter, compiled
caller or reference run a test wants is started by THIS process on request, so that no fork() / vfork() ever happens in the pytest process once it holds
GPU mappings (fork() there starves concurrent copies, DESIGN_LOG.md section 8, and was the one thing in front of both unexplained aborts of round 5).
Protocol: one JSON object per line on stdin {argv, env, cwd, timeout, input} -> one per line on stdout {returncode, stdout, stderr, timeout}.
It never imports torch and never touches a GPU itself; it ends when its stdin closes."""
import json, subprocess, sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            req = json.loads(line)
            try:
                r = subprocess.run(req["argv"], env=req.get("env"), cwd=req.get("cwd"), capture_output=True, text=True, errors="replace",
                                   timeout=req.get("timeout"), input=req.get("input"))
                resp = {"returncode": r.returncode, "stdout": r.stdout, "stderr": r.stderr, "timeout": False}
            except subprocess.TimeoutExpired as e:
                dec = lambda b: b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
                resp = {"returncode": -9, "stdout": dec(e.stdout), "stderr": dec(e.stderr), "timeout": True}
        except Exception as e:      # noqa: BLE001
            resp = {"returncode": 127, "stdout": "", "stderr": "childserver: %r" % (e,), "timeout": False}
        sys.stdout.write(json.dumps(resp) + "\n"); sys.stdout.flush()


if __name__ == "__main__":
    main()
