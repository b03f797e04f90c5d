"""Configuration surface of the PCAA path.

Mirrors the names the reference's model/loop code reads from its ``constants``
module (reference ``constants.py:1-97``): the model code reads
``NFEATURES/NSTEPS/NMAX/POINTNET_OUT_DIM/DTC_FILTERS/SUP_LATENT_DIM`` at module
construction time (``models.py:87-97``, ``112-149``, ``249-260``, ``344``) and the
loops read the ``CONFIG`` dict keys (``PCAA_ablation.py:751-1087``).

``DEVICE`` differs on purpose: this package is MI355X-only, the product path has
no CPU fallback.  ``DEVICE`` is "cuda" (= HIP device on PyTorch-ROCm) always;
importing the package on a box without a GPU is fine, running a module is not.
"""
import os
from enum import Enum


class SPLIT(Enum):
    TRAIN = "train"
    VALID = "valid"
    TEST = "test"
    UNSEEN = "unseen"


class SCENARIO(Enum):
    FREE_WALK = "free_walk"
    HANDS_IN_POCKETS = "hands_in_pockets"
    SMARTPHONE = "smartphone"


DATA_PATH = os.path.join("..", "..", "radar_reid_pytorch", "data", "multi-scenario_dataset")
GEN_DATA_PATH = os.path.join("data", "generated_dataset")

DEVICE = "cuda"

# geometry of one crop (reference constants.py:29-32)
NMAX = 150
NSTEPS = 30
CROP_STEP = 6
NFEATURES = 4

# network widths (reference constants.py:36-39)
POINTNET_OUT_DIM = 1024
DTC_FILTERS = [16, 32, 64, 128, 256, 512]
SUP_LATENT_DIM = 32
DEC_MLP_SIZE = NSTEPS * NMAX * NFEATURES

# optimiser (reference constants.py:44-48)
LR = 1e-4
B1 = 0.9
B2 = 0.99

TRAIN_CLASSES = []
TRAIN_SCENARIOS = [SCENARIO.FREE_WALK, SCENARIO.HANDS_IN_POCKETS, SCENARIO.SMARTPHONE]

BATCH_SIZE = 16
SUBSAMPLE_FACTOR = 1.0
EPOCHS = 50
CHECKPOINT_FREQUENCY = 5
GP_WEIGHT = 15
ADV_WEIGHT = 1

WANDB_PROJECT = "PCAA"
WANDB_MODE = "disabled"
MODEL_NAME = ""
NOTES = ""
SUPERVISION_FREQUENCY = 1

# keys the training/inference loops read from the config dict
# (reference constants.py:73-97; uses at PCAA_ablation.py:751-1087)
_CONFIG_KEYS = (
    "NMAX NSTEPS CROP_STEP NFEATURES POINTNET_OUT_DIM DTC_FILTERS SUP_LATENT_DIM "
    "DEC_MLP_SIZE LR B1 B2 TRAIN_CLASSES TRAIN_SCENARIOS SUBSAMPLE_FACTOR EPOCHS "
    "BATCH_SIZE GP_WEIGHT ADV_WEIGHT MODEL_NAME NOTES CHECKPOINT_FREQUENCY "
    "SUPERVISION_FREQUENCY"
).split()
CONFIG = {k: globals()[k] for k in _CONFIG_KEYS}
