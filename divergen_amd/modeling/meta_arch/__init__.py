from .custom_rcnn import CustomRCNN  # noqa
