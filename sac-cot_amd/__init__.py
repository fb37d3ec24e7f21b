"""saccot_amd — MI355X-native SAC-COT compatibility-triangle sample-consensus hot path.

The directory is named `sac-cot_amd` (not importable by name); load it with `__graft_entry__.load_package()`,
which registers it as the module `saccot_amd`.
"""
from . import api, corrio, shard, synth  # noqa: F401
from .api import (MultiRegistrar, Registrar, SacCotError, ScParams, ScStats, ScShardPlan, load_library, make_params, register, shard_plan,  # noqa: F401
                  SC_AOS, SC_SOA, SC_RANK_WEIGHT, SC_RANK_DEGREE, SC_SCORE_COUNT, SC_SCORE_MSE, SC_SCORE_MAE, SC_FLAG_TIMING, SC_FLAG_EXACT_TOTAL, SC_FLAG_NO_PRUNE, SC_FLAG_REFINE, SC_FLAG_TIMING_HOT, SC_FLAG_NO_DENSE_S, SC_FLAG_TIMING_ONE, SC_TIMING_STAGE, SC_HIST_WORDS, SC_OK, SC_ENOHYP, SC_EINVAL, SC_ERETRY, SC_EBOUND, SC_FLAG_EST_BOUND, SC_FLAG_SHARD_AB)

__all__ = ["api", "corrio", "shard", "synth", "Registrar", "SacCotError", "ScParams", "ScStats", "load_library", "make_params",
           "register"]
