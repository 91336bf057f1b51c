"""ctypes binding of libdb1_hip.so.  The prototypes are read from include/db1_hip.h so the
Python side can never drift from the C ABI.  There is NO fallback: if the library is missing or a
call fails, this raises (the product path must fail loudly, never route through a CPU path).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "db1_hip.h")
TEST_HEADER = os.path.join(HERE, "..", "include", "db1_hip_test.h")   # thread-local kernel-steering hooks of the tests / tuning tools
LIB_PATH = os.path.join(HERE, "libdb1_hip.so")

DB1_F32, DB1_BF16 = 0, 1
ACT_CODES = {"geglu": 0, "gelu": 1, "relu": 2}

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32,
    "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32,
    "void": None,
}


class Db1Error(RuntimeError):
    pass


def parse_header(path: str = HEADER, with_test_hooks: bool = False) -> Dict[str, Tuple[object, List[object]]]:
    """Returns {name: (restype, [argtypes])} for every function declared in the header (``with_test_hooks``: and in db1_hip_test.h)."""
    src = open(path).read()
    if with_test_hooks:
        src += open(TEST_HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*(const\s+char\s*\*|int64_t|int|void)\s+(db1_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else (None if ret == "void" else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int))
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPES[base])
        protos[name] = (restype, argtypes)
    return protos


_lib = None
_protos = None


def load():
    """Load the shared library (build it first with bdm_db1_amd.build.build_lib())."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    # torch must be imported first: it loads the HIP runtime (libamdhip64) this process will use for its
    # tensors and streams, and libdb1_hip.so has to bind to that SAME runtime instance.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise Db1Error(f"{LIB_PATH} is missing: run `python -m bdm_db1_amd.build` (or __graft_entry__.build()). "
                       "There is no CPU fallback for the DB1 hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header(with_test_hooks=True)
    for name, (restype, argtypes) in _protos.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.db1_is_experiment_build() and os.environ.get("DB1_ALLOW_EXPERIMENT") != "1":
        raise Db1Error(f"{LIB_PATH} was compiled with -DDB1_EXPERIMENT (timing ablations with wrong results by construction, tools/exp): "
                       "rebuild without DB1_EXTRA_HIPCC_FLAGS, or set DB1_ALLOW_EXPERIMENT=1 in the experiment script")
    _lib = lib
    return lib


# A/B knobs of the dispatchers (include/db1_hip_test.h: db1_test_set_knob; thread-local in the library, never read from the environment by
# the library itself).  Tuning scripts under tools/ keep their old switches by calling this once: DB1_W4=0 python tools/exp/exp_w4.py ...
KNOB_ENV = {"DB1_GEMM_TILE": "gemm_tile", "DB1_GEMM_SPLITK": "gemm_splitk", "DB1_GEMM_PP32_STAGES": "pp32_stages",
            "DB1_LINEAR_DECODE_SPLITK": "linear_decode_splitk", "DB1_W4": "w4", "DB1_FLASH_FWD2": "flash_fwd2", "DB1_FLASH_KV3": "flash_kv3",
            "DB1_CONV_WGRAD_KS": "conv_wgrad_ks", "DB1_GEGLU_EPI": "geglu_epi", "DB1_GEMM_HALFWAVE": "gemm_halfwave", "DB1_W4N": "w4n", "DB1_CONV_PATCH": "conv_patch", "DB1_TRI_SPLIT": "tri_split"}


def set_knob(name: str, value: int):
    call("db1_test_set_knob", name.encode(), int(value))


def apply_env_knobs():
    """(tools / tests only) forward the DB1_* A/B environment switches of the calling script to the CURRENT thread's knobs"""
    for env, knob in KNOB_ENV.items():
        if env in os.environ:
            set_knob(knob, int(os.environ[env]))


def declared_symbols() -> List[str]:
    return sorted(parse_header().keys())


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().db1_last_error()
        raise Db1Error(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    fn = getattr(load(), name)
    st = fn(*args)
    if st != 0:
        msg = _lib.db1_last_error()
        raise Db1Error(f"{name} failed with status {st}: {msg.decode() if msg else ''}")
