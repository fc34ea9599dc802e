"""elektronn2_amd -- MI355X-native hot path behind ELEKTRONN2's neuromancer
Conv / Pool / UpConv node API.  HIP-only: importing
``elektronn2_amd.backend`` without ``libe2hip.so`` raises."""
__version__ = "0.1.0"
