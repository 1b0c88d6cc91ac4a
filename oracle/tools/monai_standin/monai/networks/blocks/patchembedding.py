from networks.blocks.patch_embedding import PatchEmbeddingBlock  # reference's vendored copy  # noqa: F401
