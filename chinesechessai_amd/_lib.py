"""ctypes binding of libxq_hip.so (include/xq_selfplay.h).

The HIP library is the product: there is NO CPU fallback.  If the shared object is missing or no
gfx950 device is visible, every compute entry point raises — loudly.
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libxq_hip.so")
SOURCES = [os.path.join(CSRC, "xq_engine.hip"), os.path.join(CSRC, "xq_conv.hip"), os.path.join(CSRC, "xq_replay.hip"),
           os.path.join(CSRC, "xq_tower.hip"), os.path.join(CSRC, "xq_policy.hip")]
HEADERS = [os.path.join(CSRC, "xq_device.hpp"), os.path.join(CSRC, "xq_attack.hpp"), os.path.join(CSRC, "xq_mfma.hpp"), os.path.join(CSRC, "xq_tower_probes.hpp"),
           os.path.join(CSRC, "xq_tower1wa.hpp"), os.path.join(CSRC, "xq_tower1wa_body.inc"), os.path.join(CSRC, "xq_policy_fc1w_body.inc"),
           os.path.join(_HERE, "..", "include", "xq_selfplay.h"), os.path.join(_HERE, "..", "include", "xq_debug.h")]

MAX_MOVES = 128
MAX_PLIES = 70
WINNER_NONE = 2
NO_KING = -1
POLICY_SIZE = 8100
STATE_WORDS = 10
(S_PLAYER, S_MOVE_COUNT, S_WINNER, S_RED_KING, S_BLACK_KING, S_NO_CAPTURE, S_CONSEC_CHECKS, S_REASON,
 S_REASON_SIDE, S_REASON_COUNT) = range(10)
EVAL_PRIORS, EVAL_LOGITS_F32, EVAL_LOGITS_BF16 = 0, 1, 2
PLANES_NONE, PLANES_NCHW_F32, PLANES_NCHW_BF16, PLANES_NHWC16_BF16 = 0, 1, 2, 3
SAMPLE_RECORD_BYTES = 48 + 8 + 4 + 4 + 256 + 256


class XqError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("n_games", C.c_int32), ("sims", C.c_int32), ("leaf_batch", C.c_int32),
                ("max_moves", C.c_int32), ("temperature", C.c_double), ("opponent_mode", C.c_int32),
                ("planes_format", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32)]


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def build(force=False, verbose=False):
    """Cross-compile the HIP library for gfx950 in-tree (works without a GPU)."""
    headers = list(HEADERS)
    # XQ_TOWER_PROBES=1 in the environment also compiles csrc/xq_tower_probes.hpp: the trunk kernel's experiments and timing
    # probes (tools/bench_tower.py, tools/probe_tiles.py, tools/probe_loop.py); the default library leaves them out
    probes = os.environ.get("XQ_TOWER_PROBES", "0") == "1"
    if probes:
        # the timing-only bodies of the two generated asm statements are not committed: written when missing or older than
        # their generator, and dependencies of the objects like every other header
        tools = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "tools")
        for gen, inc in (("gen_tower1wa.py", "xq_tower1wa_abl.inc"), ("gen_policy_fc1w.py", "xq_policy_fc1w_abl.inc")):
            gp, ip = os.path.join(tools, gen), os.path.join(CSRC, inc)
            if not os.path.exists(ip) or os.path.getmtime(ip) < os.path.getmtime(gp):
                subprocess.check_call([sys.executable, gp, "--ablations"], stdout=subprocess.DEVNULL)
            headers.append(ip)
    deps = SOURCES + headers
    flagfile = os.path.join(CSRC, "build", "flags.txt")
    built_with = open(flagfile).read().strip() if os.path.exists(flagfile) else ""
    if probes and built_with != "probes":
        force = True                      # (without the request any up-to-date library will do: a probes build is a superset)
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    # one hipcc per source, concurrently (the trunk kernel alone takes ~2 min), then one link
    flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    if probes:
        flags.append("-DXQ_TOWER_PROBES=1")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs, tower_rebuilt = [], False
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in [src] + headers):
            tower_rebuilt = tower_rebuilt or src.endswith("xq_tower.hip")
            cmd = [hipcc_path()] + flags + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + [
        os.path.join(objdir, os.path.basename(src) + ".o") for src in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(flagfile, "w") as f:
        f.write(("probes" if probes or (built_with == "probes" and not tower_rebuilt) else "default") + "\n")
    return LIB_PATH


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so.  Two HIP
    runtimes in one process cannot both own the GPU, so when torch is installed its runtime is
    loaded first and libxq_hip.so (NEEDED libamdhip64.so.7) binds to that same copy."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamd_comgr.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


_I8P, _I32P, _U8P, _U16P, _U32P, _U64P, _F32P, _F64P = (C.POINTER(t) for t in (
    C.c_int8, C.c_int32, C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_float, C.c_double))

_SIGNATURES = {
    "xq_last_error": (C.c_char_p, []),
    "xq_device_count": (C.c_int, []),
    "xq_device_ok": (C.c_int, [C.c_int]),
    "xq_rules_legal_moves": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_rules_query": (C.c_int, [C.c_int] + [C.c_void_p] * 7),
    "xq_rules_threatened_pieces": (C.c_int, [C.c_int] + [C.c_void_p] * 6),
    "xq_rules_position_key": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_rules_make_move": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "xq_engine_destroy": (None, [C.c_void_p]),
    "xq_engine_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_set_pow_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "xq_engine_new_games": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_set_logit_columns": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "xq_engine_set_temperature": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_int]),
    "xq_engine_set_root_noise": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_uint64]),
    "xq_engine_set_tree_reuse": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_set_virtual_loss": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_leaf_slots": (C.c_int, [C.c_void_p]),
    "xq_engine_tree_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_engine_read_root_priors": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_set_uniforms": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_set_roots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_engine_search_round": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_engine_eval_hashnet": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_end_search": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "xq_engine_play_move": (C.c_int, [C.c_void_p]),
    "xq_engine_finalize": (C.c_int, [C.c_void_p]),
    "xq_engine_active_games": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_active_games_post": (C.c_int, [C.c_void_p]),
    "xq_engine_active_games_poll": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_engine_priors_ptr": (C.c_void_p, [C.c_void_p]),
    "xq_engine_values_ptr": (C.c_void_p, [C.c_void_p]),
    "xq_engine_rounds_per_move": (C.c_int, [C.c_void_p]),
    "xq_engine_read_leaves": (C.c_int, [C.c_void_p] * 6),
    "xq_engine_write_priors": (C.c_int, [C.c_void_p] * 3),
    "xq_engine_read_root_visits": (C.c_int, [C.c_void_p] * 4),
    "xq_engine_read_tree": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8),
    "xq_engine_read_games": (C.c_int, [C.c_void_p] * 8),
    "xq_engine_read_samples": (C.c_int, [C.c_void_p] * 9),
    "xq_engine_pack_samples": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_set_root_eval_carry": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_roots_not_ready": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_set_row_compaction": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_set_leaf_dedupe": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_set_eval_cache": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_eval_cache_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "xq_engine_row_map": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "xq_engine_read_row_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "xq_engine_read_leaf_rows": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_engine_refill_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "xq_engine_refill_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "xq_engine_refill_read_games": (C.c_int, [C.c_void_p] * 8),
    "xq_engine_refill_read_slots": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xq_conv3x3_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int]),
    "xq_heads_nhwc_bf16": (C.c_int, [C.c_void_p] * 6 + [C.c_int]),
    "xq_tower_nhwc_bf16": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "xq_policy_fc_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "xq_value_head_bf16": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_void_p]),
    "xq_replay_last_error": (C.c_char_p, []),
    "xq_replay_create": (C.c_int, [C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "xq_replay_destroy": (None, [C.c_void_p]),
    "xq_replay_size": (C.c_int64, [C.c_void_p]),
    "xq_replay_push_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "xq_replay_encode_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "xq_replay_read_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "xq_engine_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "xq_engine_profile_read": (C.c_int, [C.c_void_p] * 5),
}

# diagnostics (include/xq_debug.h): build selectors, phase stamps, timing probes - tools and build-comparison tests only
_DEBUG_SIGNATURES = {
    "xq_tower_set_variant": (None, [C.c_int]),
    "xq_tower_set_clock_sample": (None, [C.c_void_p]),
    "xq_tower_debug_stamps": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]),
    "xq_mfma_probe": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int]),
    "xq_conv3x3_set_variant": (None, [C.c_int]),
    "xq_conv3x3_debug_stamps": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    "xq_engine_set_search_occupancy": (None, [C.c_int]),
    "xq_engine_set_search_stamps": (C.c_int, [C.c_void_p]),
    "xq_policy_fc_set_variant": (None, [C.c_int]),
    "xq_policy_fc_debug_stamps": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "xq_policy_fc_debug": (C.c_int, [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int]),
}

EXPORTS = sorted(_SIGNATURES)
DEBUG_EXPORTS = sorted(_DEBUG_SIGNATURES)
ROW_HISTORY = 65536


def lib():
    """Load libxq_hip.so; raises XqError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # a fresh checkout: compile in-tree if the toolchain is here (seconds); never fall back to CPU
            try:
                build()
            except Exception as ex:
                raise XqError("libxq_hip.so is not built (%s) and could not be compiled (%s): run "
                              "`python -c 'import __graft_entry__ as g; g.build()'`; the HIP path has no CPU fallback"
                              % (LIB_PATH, ex))
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for table in (_SIGNATURES, _DEBUG_SIGNATURES):
            for name, (res, args) in table.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().xq_last_error()
        raise XqError("libxq_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(a):
    """numpy array -> void* (None passes NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
