"""Mirror of the reference package ``pretrain_src/model`` (vilmodel, pretrain_cmt)."""
