from .init_embeddings import ScaledEmbedding, ZeroEmbedding  # noqa: F401
