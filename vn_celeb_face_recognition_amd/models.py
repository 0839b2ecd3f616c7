"""Plugin registry mirroring /root/reference/models/__init__.py:1-7: the CLIs resolve detector,
encoder and classifier classes with getattr(models, <name>)(**json_kwargs)
(demo_image.py:361-374, demo_video.py:260-273, find_embedding.py:77)."""
from .encoders import InceptionResnetV1, iresnet100  # noqa: F401
from .classifier import MLPModel  # noqa: F401
from .detector import MTCNN  # noqa: F401
from .retina import RetinaFace  # noqa: F401


def _out_of_scope(name, why):
    def ctor(*a, **k):
        raise NotImplementedError("%s is outside the MI355X hot path (%s); see SURVEY.md section 8" % (name, why))
    ctor.__name__ = name
    return ctor


resnet101 = _out_of_scope("resnet101", "alternative encoder, weights not shipped")
resnet_2branch_50 = _out_of_scope("resnet_2branch_50", "emotion recognition")
