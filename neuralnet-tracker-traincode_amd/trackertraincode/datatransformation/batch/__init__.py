from .geometric import (GeneralFocusRoi, MakeRoiRandomizationParameters, NoRoiRandomization,  # noqa: F401
                        RoiFocusRandomizationParameters)
