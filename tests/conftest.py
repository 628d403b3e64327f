import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """libza_amd.so and the oracle library, built in-tree (hipcc cross-compiles without a GPU)."""
    from za_amd import build as zbuild
    zbuild.build()
    from oracle import coracle
    coracle.build()
    import za_amd
    return za_amd.load_library()


@pytest.fixture(scope="session")
def ctx(lib_built):
    import za_amd
    c = za_amd.Context(0)   # raises ZaAmdError(ZG_E_NO_DEVICE) without a GPU: no CPU fallback
    yield c
    c.close()


@pytest.fixture
def opts(ctx):
    """zg_set_option on the shared context for one test: opts(za_amd.OPT_TABLES, 0); the defaults come back afterwards."""
    import za_amd
    defaults = {za_amd.OPT_PACK: 1, za_amd.OPT_PACK_THREADS: 0, za_amd.OPT_PACK_MIN_LOG_N: 18, za_amd.OPT_PACK_PIN: 1,
                za_amd.OPT_H_EVAL: 1, za_amd.OPT_TABLES: 31, za_amd.OPT_TABLES_MAX_MB: 0, za_amd.OPT_PACK_BUDGET_PERMILLE: 900,
                za_amd.OPT_PACK_RAW_PERMILLE: 400, za_amd.OPT_COMBINE_EARLY: 1, za_amd.OPT_SMALL_LOG_N: 16,
                za_amd.OPT_SMALL_W_LOG_N: -1, za_amd.OPT_SORT_STAGED: -1, za_amd.OPT_PART_STAGE_MIN: -1, za_amd.OPT_SUBSET_W: 8,
                za_amd.OPT_COMPACT_TABLE: 1, za_amd.OPT_H_PARTS: 0, za_amd.OPT_H_PIPE: -1, za_amd.OPT_SHARE_SORT: 1, za_amd.OPT_NTT_BATCH_LOG_M: 21,
                za_amd.OPT_C_FOLD: 1, za_amd.OPT_C_FOLD_MIN_LOG_M: 20, za_amd.OPT_UPLOAD_GATES: 23}
    # (a whole-suite run with a schedule forced through ZA_AMD_OPTIONS keeps it after a test that touched the option)
    defaults.update(dict(za_amd._options_from_env()))
    touched = set()

    def set_(option, value):
        touched.add(option)
        ctx.set_option(option, value)

    yield set_
    for o in touched:
        ctx.set_option(o, defaults[o])
