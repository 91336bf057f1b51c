from .input_specs import GatoInputBase, RLTaskInput, NLPTaskInput, ICTaskInput, VQATaskInput  # noqa: F401
