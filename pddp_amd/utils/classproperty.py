"""`classproperty` descriptor (reference: pddp/utils/classproperty.py)."""


class classproperty(object):

    def __init__(self, fget):
        self.fget = fget

    def __get__(self, instance, owner):
        return self.fget(owner)
