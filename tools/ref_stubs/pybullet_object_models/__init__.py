class _M(object):
    @staticmethod
    def getDataPath():
        return "/nonexistent"
ycb_objects = _M()
superquadric_objects = _M()
