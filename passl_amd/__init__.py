"""passl_amd — MI355X-native (gfx950) hot paths of PaddlePaddle/PASSL behind the reference's Python interface."""
