from .project import Project  # noqa: F401
