from .sam import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam  # noqa: F401
