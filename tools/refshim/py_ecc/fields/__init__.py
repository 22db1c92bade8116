from .field_elements import FQ
