from speech_decoding_amd.layout import ch_locations_2d  # noqa: F401
