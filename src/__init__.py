"""Drop-in import paths of the reference package (`src.*`) resolved onto the MI355X implementation in `oneprot_amd`."""
