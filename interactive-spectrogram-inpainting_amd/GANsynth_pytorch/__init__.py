"""MI355X build of the part of `GANsynth_pytorch` the reference calls (the package itself is absent
from the reference tree and un-pinned): `spectrograms_helper.{SpectrogramsHelper, MelSpectrogramsHelper}`."""
