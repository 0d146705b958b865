"""wesep_amd: MI355X-native pBSRNN target-speaker-extraction training path (drop-in for the
hot path of wenet-e2e/wesep).  Importing the package does not load the HIP library; the first
op does, and raises if libwesep_hip.so is missing (there is no CPU fallback)."""
__version__ = "0.1.0"
