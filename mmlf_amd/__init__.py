"""mmlf_amd: MI355X-native EPI-stack CNN forward/backward path of titus-leistner/mmlf.

Public surface mirrors the reference's model package for this path
(reference mmlf/model/feed_forward.py, ensamble.py, loss.py; mmlf/utils/dl.py).
"""
__all__ = ['FeedForward', 'Ensamble']


def __getattr__(name):
    if name == 'FeedForward':
        from .feed_forward import FeedForward
        return FeedForward
    if name == 'Ensamble':
        from .ensamble import Ensamble
        return Ensamble
    raise AttributeError(name)
