"""Independent fp32/fp64 forward of a parsed Keras graph on torch-CPU (test helper) -- used to
cross-check oracle/keras_forward.py (SURVEY.md 8c "substitute evidence" ii)."""
from tools.synth_model import forward_torch  # noqa: F401
