"""MI355X-native trajectory-analysis engine behind the MDSuite calculator API."""
__version__ = "0.1.0"
