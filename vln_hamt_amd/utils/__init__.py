"""Mirror of the one piece of ``pretrain_src/utils`` that sits on the hot path's multi-GPU boundary: ``misc.wrap_model``."""
