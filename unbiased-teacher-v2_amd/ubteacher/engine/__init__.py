from .trainer import UBTeacherTrainer, UBRCNNTeacherTrainer  # noqa: F401  (reference engine/__init__.py:3)
