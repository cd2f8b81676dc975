"""sbb_textline_detection_amd -- MI355X-native drop-in for the patch-wise segmentation inference
path of qurator-spk/sbb_textline_detection (``do_prediction`` / ``model.predict``)."""
from .keras_graph import parse_model_config, resnet50_unet_config  # noqa: F401
from .model import (SegModel, Session, clear_session, load_model,  # noqa: F401
                    start_new_session_and_model)
from .predict import PatchSegmenter, do_prediction, do_prediction_pages, resize_nearest  # noqa: F401
from .stages import InferenceStages, otsu_copy, scaled_size  # noqa: F401
from .weights import load_sbbw, save_sbbw, synthetic_model  # noqa: F401

__version__ = "0.1.0"
