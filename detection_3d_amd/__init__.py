"""MI355X-native hot path of zhupan007/Detection_3D."""
__version__ = "0.1.0"
