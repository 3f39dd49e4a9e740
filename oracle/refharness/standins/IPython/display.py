def display(*objs, **kwargs):
    for o in objs:
        print(o)
