"""Mirror of the reference package ``finetune_src/models`` (vilmodel_cmt, model_HAMT, vlnbert_init)."""
