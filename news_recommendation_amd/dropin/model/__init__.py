"""Drop-in ``model`` package: same import paths, class names, constructor / forward signatures and
state_dict keys as the reference's ``src/model`` tree, with the math running in the
MI355X HIP kernels.  Put ``news_recommendation_amd/dropin`` ahead of the reference's ``src`` on sys.path
(news_recommendation_amd/launcher.py does) and the reference's train.py / evaluate.py run unchanged.
The hot-path models of SURVEY.md section 8 are provided: NRMS, NAML, LSTUR."""
