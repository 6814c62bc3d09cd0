"""HIP sources of libpassl_hip.so and the build script (python -m passl_amd.csrc.build)."""
