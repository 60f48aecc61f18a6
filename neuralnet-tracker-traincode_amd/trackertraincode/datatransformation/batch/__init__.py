from .geometric import (GeneralFocusRoi, MakeRoiRandomizationParameters, NoRoiRandomization,  # noqa: F401
                        RoiFocusRandomizationParameters)
from .intensity import (KorniaImageDistortions, OnlyClip, RandomBrightness, RandomContrast, RandomEqualize,  # noqa: F401
                        RandomGamma, RandomGaussianBlur, RandomGaussianNoise, RandomGaussianNoiseWithClipping, RandomPosterize)
