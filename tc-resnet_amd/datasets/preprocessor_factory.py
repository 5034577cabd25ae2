"""Preprocessor registry, same keys as the reference (datasets/preprocessor_factory.py:6-19)."""
from .preprocessors import LogMelSpectrogramPreprocessor, MFCCPreprocessor, NoOpPreprocessor

_available_preprocessors = {
    "log_mel_spectrogram": LogMelSpectrogramPreprocessor,
    "mfcc": MFCCPreprocessor,
    "no_preprocessing": NoOpPreprocessor,
}


def factory(preprocess_method, scope, preprocessed_node_name):
    if preprocess_method in _available_preprocessors:
        return _available_preprocessors[preprocess_method](scope, preprocessed_node_name)
    raise NotImplementedError(f"{preprocess_method}")
