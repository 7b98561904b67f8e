"""focoos_amd — MI355X-native engine for the focoos RT-DETR hot path."""
__version__ = "0.1.0"
