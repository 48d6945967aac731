"""zero-tig_amd: MI355X-native (gfx950) implementation of the Zero-TIG training / inference hot path."""
