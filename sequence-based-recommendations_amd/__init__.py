"""MI355X-native engine for the `train.py -m RNN` hot path (import as `sbr_amd`)."""
