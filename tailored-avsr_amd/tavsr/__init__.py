"""tavsr: MI355X-native host mirror of the tailored-avsr hot path."""
__version__ = "0.1.0"
