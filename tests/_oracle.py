"""Test-side loader of the CPU oracle (oracle/libmpc_oracle.so).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may use this; the product package never imports it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libmpc_oracle.so")

_lib = None


def host_stamp():
    """CPU model + a digest of its ISA flags: the libraries are compiled -march=native, so a build from another host is rebuilt here."""
    import hashlib
    model, flags = "", ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if not model and line.startswith("model name"):
                    model = line.strip()
                if not flags and line.startswith("flags"):
                    flags = line
                if model and flags:
                    break
    except OSError:
        pass
    return model + "\n" + hashlib.md5(flags.encode()).hexdigest() + "  -\n"


def built_on(lib_dir):
    """-> (stamp text of the host that built the library or None, True if that is this host)"""
    p = os.path.join(lib_dir, ".build_host")
    if not os.path.exists(p):
        return None, False
    with open(p) as f:
        txt = f.read()
    return txt, txt == host_stamp()


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    global _lib
    if _lib is None and os.environ.get("MPC_ORACLE_LIBRARY"):  # e.g. the sanitizer build (oracle/Makefile: asan-test)
        from mpc_benchmark_amd import _capi
        _lib = _capi.bind_library(os.environ["MPC_ORACLE_LIBRARY"])
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
        stale = (not os.path.exists(ORACLE_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs)
        if stale:
            try:
                build()
            except Exception:
                if not os.path.exists(ORACLE_LIB):
                    raise
        from mpc_benchmark_amd import _capi
        _lib = _capi.bind_library(ORACLE_LIB)
    return _lib
