"""MI355X-native NRMS scoring engine (hand-written HIP kernels behind the
reference's NewsEncoder / UserEncoder / click_predictor module API)."""
__version__ = "0.1.0"
