"""MI355X-native implementation of the neural-motifs hot path behind the reference's `lib.*` import surface."""
