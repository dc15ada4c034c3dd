"""srfrd_amd: the SRFRD sequential-recommender hot path on MI355X (gfx950).

Drop-in module classes (same constructors / forward / predict / state_dict as the reference's SRFR_model.py), a fused
train step, batched evaluation and a synthetic sampler, all on top of the C ABI in include/srfrd_hip.h.
"""
from .modules import SASRec, SRFR, SRFRN, SRFU, SRFU_B, SRFU_F, SRFU_R  # noqa: F401
from .trainer import FusedTrainer, flat_allreduce, shard_bounds  # noqa: F401
from .evaluate import evaluate_batches, ranks_from_logits  # noqa: F401
from .sampler import synthetic_batch, eval_candidates  # noqa: F401
from .ranker import ShardedRanker, row_shards, topk_merge  # noqa: F401
from .exchange import GradExchange  # noqa: F401
from .optim import Adam  # noqa: F401
from .dataset import (InteractionData, partition, load_csv, eval_inputs, evaluation, DeviceSampler,  # noqa: F401
                      load_reference_checkpoint)

__all__ = ["SASRec", "SRFR", "SRFRN", "SRFU", "SRFU_B", "SRFU_F", "SRFU_R", "FusedTrainer", "flat_allreduce",
           "shard_bounds", "evaluate_batches", "ranks_from_logits", "synthetic_batch", "eval_candidates", "InteractionData",
           "partition", "load_csv", "eval_inputs", "evaluation", "DeviceSampler", "load_reference_checkpoint", "ShardedRanker",
           "row_shards", "topk_merge", "GradExchange", "Adam"]
