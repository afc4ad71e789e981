"""A/B harness of tools/: the kernel-form switches of a run are given as DYF_* variables in the TOOL's environment
(`DYF_HALO_SPLITK=0 python tools/bench_small_rows.py ...`), read HERE -- by the tool -- and handed to the library through
dyf_debug_set_form (include/dyffusion_hip_testing.h).  libdyffusion_hip.so itself reads no kernel-form switch from the
environment; a process that does not import this module runs the production policy whatever its environment holds."""
import os

# variables that are not kernel-form switches: the library's two user-facing ones, the python binding's, bench.py's, the tools' own
NOT_FORMS = ("DYF_VERBOSE", "DYF_RCCL_LIB", "DYF_LIB", "DYF_LIB_F16", "DYF_BENCH_", "DYF_DIST_BACKEND", "DYF_CPU_THREADS",
             "DYF_ALLOW_BF16_LONG_ROLLOUT", "DYF_PMC_REGEX", "DYF_OISST_DTYPE", "DYF_NO_GRAPH", "DYF_SMALL_MAXB", "DYF_ORACLE_CACHE",
             "DYF_WRITE_ORACLE_CACHE", "DYF_TEST_FORMS", "DYF_EXPERIMENT_BUILD")


def forward_env_forms(verbose=True):
    from dyffusion_amd import _lib

    set_ = {}
    for key, value in sorted(os.environ.items()):
        if key.startswith("DYF_") and not key.startswith(NOT_FORMS):
            _lib.set_form(key, value)
            set_[key] = value
    if set_ and verbose:
        print(f"[tools/_forms] kernel-form switches handed to dyf_debug_set_form: {set_}", flush=True)
    return set_
