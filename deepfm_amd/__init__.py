"""deepfm_amd — MI355X-native CTR feature-interaction path (drop-in for CodexploreRepo/deepfm's
FeatureEmbedding / FMInteraction / CIN / MultiHeadSelfAttention and the models built on them)."""

__version__ = "0.1.0"
