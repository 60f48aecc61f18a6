from .geometric import (GeneralFocusRoi, MakeRoiRandomizationParameters, NoRoiRandomization,  # noqa: F401
                        RoiFocusRandomizationParameters)
from .intensity import (KorniaImageDistortions, OnlyClip, RandomBrightness, RandomContrast, RandomEqualize,  # noqa: F401
                        RandomGamma, RandomGaussianBlur, RandomGaussianNoise, RandomGaussianNoiseWithClipping, RandomPosterize)
from .misc import PutRoiFromLandmarks, head_extent_roi  # noqa: F401
