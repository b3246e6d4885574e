"""keypointnerf_amd — MI355X-native (gfx950) ray-march renderer for KeypointNeRF's hot path."""
__version__ = "0.1.0"
