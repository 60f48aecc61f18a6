"""MI355X-native drop-in for the pose-estimator training path of opentrack/neuralnet-tracker-traincode.

Same module layout and public names as the reference's `trackertraincode` package for the hot path
(backbones / neuralnets / train / pipelines surface); the arithmetic runs in hand-written HIP
kernels (../csrc) reached through the C-ABI in include/ttk.h.
"""
