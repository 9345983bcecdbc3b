"""vita_amd - MI355X-native kernels and drop-in modules for the ViTa pretraining hot path."""
