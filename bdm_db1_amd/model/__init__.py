from .transformer_xl import TransformerXL  # noqa: F401  (same import path shape as the reference's src/model/__init__.py:14)
